// Device side of tools/probes/aql_probe.cpp: 256 work-groups of 512 threads re-read the operand
// panels of a 16x16-tile decomposition (8 panels of 16 x 272 doubles per work-group, shared along
// tile rows / columns exactly like the stage kernel's fragments) and sum them.
// Build: hipcc --genco --offload-arch=gfx950 -O3 aql_probe_kernel.hip -o aql_probe_kernel.hsaco
#include <hip/hip_runtime.h>
typedef double v2d __attribute__((ext_vector_type(2)));

template <bool SC1>
__device__ __forceinline__ v2d ld(const v2d* p) {
  v2d r;
  // plain C++ loads: the compiler must see them to keep its own vmcnt bookkeeping (an asm load's destination was
  // re-used for an address while still in flight -> fault at address 0)
  if (SC1) r = __builtin_nontemporal_load(p);
  else     r = *p;
  return r;
}

template <bool SC1, int NL>
__device__ __forceinline__ double sum_panel(const v2d* p, int lane) {
  // NL of the 34 dwordx4 per lane that make one 16 x 272 panel, all in flight at once
  v2d acc = {0.0, 0.0};
  v2d r[NL > 0 ? NL : 1];
#pragma unroll
  for (int i = 0; i < NL; ++i) r[i] = ld<SC1>(p + i * 64 + lane);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < NL; ++i) acc += r[i];
  return acc[0] + acc[1];
}

// The stage kernel's fragment shape: lane l = (row l&15, k-chunk l>>4) reads 32 contiguous bytes (two dwordx4) of
// each 128-byte group of a row; a wave owns one K-quarter (4 groups of 16 k) of 4 panels = 32 dwordx4 per lane.
// DEPTH 0: all 32 in flight; 1: one group (8 loads) at a time; 2: one group ahead (16 in flight)
template <int DEPTH, int PACK = 0>
__device__ __forceinline__ double sum_fragments(const double* const pan[4], int kq, int lane) {
  const int row = lane & 15, ch = lane >> 4;
  v2d acc = {0.0, 0.0};
  v2d r[32];
  const v2d* base[4];
  // PACK 0: row-major panel (row stride 272 doubles).  PACK 1/2: the panel stored as 17 blocks of 16 rows x 16 k
  // (256 doubles each), block g of the wave's quarter at (kq*4+g)*256; inside a block lane l's four doubles sit
  // at l*4 (PACK 1: 32-byte stride per lane) or as two 1-KB slabs, l*2 and 128 + l*2 (PACK 2: contiguous per instruction)
  constexpr int GS = PACK ? 128 : 8;          // v2d step between groups
  constexpr int HS = PACK == 2 ? 64 : 1;      // v2d step between a lane's two halves
#pragma unroll
  for (int a = 0; a < 4; ++a)
    base[a] = PACK == 0 ? reinterpret_cast<const v2d*>(pan[a] + row * 272 + kq * 64 + ch * 4)
            : PACK == 1 ? reinterpret_cast<const v2d*>(pan[a] + kq * 1024 + lane * 4)
                        : reinterpret_cast<const v2d*>(pan[a] + kq * 1024 + lane * 2);
  auto issue = [&](int g) {
#pragma unroll
    for (int a = 0; a < 4; ++a) { r[g * 8 + a * 2] = ld<false>(base[a] + g * GS); r[g * 8 + a * 2 + 1] = ld<false>(base[a] + g * GS + HS); }
  };
  auto eat = [&](int g) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += r[g * 8 + i];
  };
  if (DEPTH == 0) {
#pragma unroll
    for (int g = 0; g < 4; ++g) issue(g);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < 4; ++g) eat(g);
  } else if (DEPTH == 1) {
#pragma unroll
    for (int g = 0; g < 4; ++g) { issue(g); __builtin_amdgcn_sched_barrier(0); eat(g); __builtin_amdgcn_sched_barrier(0); }
  } else {
    issue(0);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (g < 3) issue(g + 1);
      __builtin_amdgcn_sched_barrier(0);
      eat(g);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  return acc[0] + acc[1];
}

// mode: bit0 sc1 loads on the state panels, bit1 one plain store per wave,
//       bits 4-5 bytes read (0: all 34 loads, 1: 17, 2: 8, 3: none), bits 8-9 sharing (0: tile pattern, 1: every
//       work-group the same 8 panels, 2: private panels per work-group),
//       bits 12-13 shape (0: contiguous panel per wave, 1-3: fragment shape with DEPTH 0-2; tile pattern only)
//       bits 14-15 packing of the fragment shapes (0 row-major, 1 packed blocks 32-byte stride, 2 packed slabs)
// Overlapped chain (mode bit 16): the launch carries no barrier bit; a work-group first waits until every
// work-group of the previous launch has arrived (8 per-XCD counters, 128 B apart, target = 256 x launches so far),
// and arrives itself at the end.  Spins are bounded: on a timeout the work-group flags out[0] and carries on.
__device__ __forceinline__ void chain_wait(unsigned int* counters, unsigned int target, double* out) {
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    for (int spin = 0; spin < (1 << 18); ++spin) {
      unsigned int v = lane < 8 ? __hip_atomic_load(counters + lane * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
      v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
      if (__shfl(v, 0) >= target) break;
      if (spin == (1 << 18) - 1 && lane == 0) out[0] = -1.0;
      __builtin_amdgcn_s_sleep(2);
    }
  }
  __syncthreads();
}
__device__ __forceinline__ void chain_arrive(unsigned int* counters) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 7;   // HW_REG_XCC_ID, 4 bits
    __hip_atomic_fetch_add(counters + xcc * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

extern "C" __global__ __launch_bounds__(512) void read_panels(const double* buf, double* out, int mode, unsigned int target,
                                                              unsigned int* counters) {
  if (mode & (1 << 16)) chain_wait(counters, target, out);
  const int b = blockIdx.x, xcd = b & 7, w = b >> 3;
  const int I = (xcd >> 1) * 4 + (w & 3), J = (xcd & 1) * 8 + (w >> 2);
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int share = (mode >> 8) & 3, shape = (mode >> 12) & 3;
  double s = 0.0;
  if (shape) {
    const int role = wv >> 2, kq = wv & 3;
    const double* pan[4];
    pan[0] = buf + (size_t)((role * 2 + 0) * 16 + I) * 4352; pan[1] = buf + (size_t)((role * 2 + 1) * 16 + I) * 4352;
    pan[2] = buf + (size_t)(64 + (role * 2 + 0) * 16 + J) * 4352; pan[3] = buf + (size_t)(64 + (role * 2 + 1) * 16 + J) * 4352;
    const int pack = (mode >> 14) & 3;
    if (pack == 0) {
      if (shape == 1) s = sum_fragments<0>(pan, kq, lane);
      else if (shape == 2) s = sum_fragments<1>(pan, kq, lane);
      else s = sum_fragments<2>(pan, kq, lane);
    } else if (pack == 1) {
      if (shape == 1) s = sum_fragments<0, 1>(pan, kq, lane); else s = sum_fragments<2, 1>(pan, kq, lane);
    } else {
      if (shape == 1) s = sum_fragments<0, 2>(pan, kq, lane); else s = sum_fragments<2, 2>(pan, kq, lane);
    }
  } else {
    int idx = wv < 4 ? wv * 16 + I : 64 + (wv - 4) * 16 + J;
    if (share == 1) idx = wv;
    if (share == 2) idx = b * 8 + wv;
    const v2d* p = reinterpret_cast<const v2d*>(buf + (size_t)idx * 4352);
    const bool state = (wv == 2 || wv == 3 || wv == 4 || wv == 5);
    const bool sc1 = (mode & 1) && state;
    switch ((mode >> 4) & 3) {
      case 0: s = sc1 ? sum_panel<true, 34>(p, lane) : sum_panel<false, 34>(p, lane); break;
      case 1: s = sc1 ? sum_panel<true, 17>(p, lane) : sum_panel<false, 17>(p, lane); break;
      case 2: s = sc1 ? sum_panel<true, 8>(p, lane) : sum_panel<false, 8>(p, lane); break;
      default: break;
    }
  }
  if (mode & 2) {
    if (lane == 0) out[4096 + b * 8 + wv] = s;   // one plain 8-byte store per wave
  }
  if (s == 123.456) out[b * 512 + threadIdx.x] = s;   // never true: keeps the loads alive
  if (mode & (1 << 16)) chain_arrive(counters);
}
