// chip_handoff_probe.hip -- what does one RK stage cost as a PHASE of a chip-wide resident-tile kernel (T x T work-groups on
// all XCDs, one per CU), i.e. csrc/ldc_xcd_kernel.inc without its one-XCD restriction?
//
// The probe is the SKELETON of such a stage, with the arithmetic of the epilogue left out:
//   every work-group (I, J) of a T x T grid (T = 16: 256 work-groups = N = 256 in the tail layout; T = 9: N = 128) per phase
//     1. waves 0-3 request one STATE PANEL each (T packed 2-KB blocks: u^T[J], v^T[J], u[I], v[I]) with L1-bypassing (sc1)
//        16-byte buffer loads -- blocks other work-groups wrote in the phase before, on whatever XCD they run;
//     2. contract it with TWO operator panels that are RESIDENT IN LDS (conflict-free ds_read_b128): 8 T f64 MFMAs per wave in
//        four independent chains (T = 16: 128 MFMAs per SIMD, the count of a plain N = 256 stage), results to LDS;
//     3. barrier, the epilogue reads its eight results and stores the tile's four packed blocks (u, u^T, v, v^T) WRITE-THROUGH
//        (sc1): 8 bytes per lane straight from registers, or 16 bytes per lane through an LDS tile;
//     4. every wave's s_waitcnt vmcnt(0), barrier, ONE lane raises this work-group's flag (sc1 store, a 128-byte line of its
//        own), wave 0 polls the flags of the 2T-1 work-groups whose tiles this one reads (row I and column J of the grid;
//        option: all T*T), barrier.
//   Every stored double carries (phase, writer); every loaded fragment is CHECKED against the writer and phase it must come
//   from (stale words are counted): the hand-over form is the guide's valid form 1 (sc1 stores, every storing wave's
//   vmcnt(0), work-group barrier, one lane's sc1 flag store; sc1 poll, work-group barrier, sc1 loads only).
// Switches take the phase apart: no MFMAs / no payload loads / no payload at all (flags only) / all flags instead of mates.
// Cycle stamps (s_memtime) per segment, summed over the timed phases, for wave 0 (polls), wave 1 (a contraction wave) and
// wave 4 (idle in a plain stage) of every work-group.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/chip_handoff_probe.hip -o tools/probes/_build/chip_handoff_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
#define GA __attribute__((address_space(1)))
#define MFMA_F64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

constexpr int kThreads = 512;
constexpr int kSeg = 9;                                  // segments between the ten stamp points of a phase
constexpr unsigned long long kSpinLimit = 20000000ull;   // 0.2 s of the 100 MHz counter

struct Args {
  int T, phases, warm;
  int do_loads, do_mfma, do_stores, st16, wait_all, patched, sleep;
  double tag0;                       // tag base of this configuration (so that an earlier run's bytes never pass the check)
  double* buf[2][4];                 // [set][u, u^T, v, v^T]: T x T packed blocks of 256 doubles
  unsigned* flags;                   // word 32 x work-group
  unsigned* giveup;
  unsigned long long* out;           // per work-group: 3 waves x kSeg segment sums, then stale words, abort, real-time ticks
};
constexpr int kOutPerWg = 3 * kSeg + 3;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const double* base) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, -1, 0x00020000);
}
// one packed fragment (lane l: 32 bytes at 32 l of the 2-KB block) by two L1-bypassing 16-byte loads
__device__ __forceinline__ v4d ld_frag_sc1(const double* arr, int blk, int lane) {
  const __amdgpu_buffer_rsrc_t r = rsrc_of(arr);
  const int off = ((blk << 8) + lane * 4) * (int)sizeof(double);
  const v2d lo = __builtin_bit_cast(v2d, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16));
  const v2d hi = __builtin_bit_cast(v2d, __builtin_amdgcn_raw_buffer_load_b128(r, off + 16, 0, 16));
  return (v4d){lo[0], lo[1], hi[0], hi[1]};
}
// an operator fragment out of LDS: stored as [half][lane] 16 bytes, so each ds_read_b128 is linear over the lanes
__device__ __forceinline__ v4d ld_frag_lds(const double* frag, int lane) {
  const v2d lo = *reinterpret_cast<const v2d*>(frag + 2 * lane);
  const v2d hi = *reinterpret_cast<const v2d*>(frag + 128 + 2 * lane);
  return (v4d){lo[0], lo[1], hi[0], hi[1]};
}
__device__ __forceinline__ int xpk(int r, int c) { return (((c >> 2) << 4) + r) * 4 + (c & 3); }
__device__ __forceinline__ void st8_sc1(double* p, double v) {
  __hip_atomic_store((GA double*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st16_sc1(double* p, double v0, double v1) {
  const v2d v = {v0, v1};
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"((GA v2d*)p), "v"(v) : "memory");
}

template <int T, bool LOADS, bool MFMA>
__global__ __launch_bounds__(kThreads, 2) void probe(const Args a) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* ops = lds;                          // 4 panels x T fragments x 256 doubles
  double* res = ops + 4 * T * 256;            // 8 result tiles in accumulator order
  double* til = res + 8 * 256;                // 2 tiles 16 x 17 (the 16-byte store path)
  __shared__ int abort_s;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bx = blockIdx.x;
  int I, J;
  if (a.patched && (T & 7) == 0) {            // each XCD (b & 7) a compact T/4 x T/2 patch of tiles, like tile_of_block
    const int xcd = bx & 7, loc = bx >> 3, pr = T >> 2, pc = T >> 1;
    const int jj = loc / pr, i = loc - jj * pr;
    I = (xcd >> 1) * pr + i; J = (xcd & 1) * pc + (I + jj) % pc;
  } else { I = bx / T; J = bx % T; }
  const int me = I * T + J;                    // this work-group's id in flags and tags
  if (tid == 0) abort_s = 0;
  for (int q = tid; q < 4 * T * 256; q += kThreads) ops[q] = 1.0 / (double)(1 + (q & 1023));
  // this thread's node of the tile (threads 0..255) and its places in the packed blocks
  const int ti = 4 * (wv & 3) + (lane >> 4), tj = lane & 15;
  const int oU = xpk(ti, tj), oT = xpk(tj, ti);
  const int blkIJ = I * T + J, blkJI = J * T + I;
  unsigned long long seg[kSeg];
#pragma unroll
  for (int k = 0; k < kSeg; ++k) seg[k] = 0;
  unsigned long long stale = 0;

  auto publish = [&](int set, double tag) {    // the tile's four blocks, write-through
    if (!a.do_stores) return;
    if (!a.st16) {
      if (tid < 256) {
        st8_sc1(a.buf[set][0] + ((size_t)blkIJ << 8) + oU, tag);
        st8_sc1(a.buf[set][1] + ((size_t)blkJI << 8) + oT, tag);
        st8_sc1(a.buf[set][2] + ((size_t)blkIJ << 8) + oU, tag);
        st8_sc1(a.buf[set][3] + ((size_t)blkJI << 8) + oT, tag);
      }
    } else {
      if (tid < 256) { til[ti * 17 + tj] = tag; til[16 * 17 + ti * 17 + tj] = tag; }
      __syncthreads();
      if (tid < 256) {
        const int hh = tid & 127, pl = hh >> 1, pr = pl & 15, pc = 4 * (pl >> 4) + 2 * (hh & 1);
        const double* t = til + (tid < 128 ? 0 : 16 * 17);
        const int e = pr * 17 + pc, eT = pc * 17 + pr;
        const int arr = tid < 128 ? 0 : 2;
        st16_sc1(a.buf[set][arr] + ((size_t)blkIJ << 8) + 2 * hh, t[e], t[e + 1]);
        st16_sc1(a.buf[set][arr + 1] + ((size_t)blkJI << 8) + 2 * hh, t[eT], t[eT + 17]);
      }
    }
  };
  // hand-over: drain, barrier, raise, wait for the mates (or all), barrier.  Stamps 6..9 of the phase.
  auto handover = [&](unsigned target, unsigned long long* P) -> bool {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (P) P[6] = __builtin_amdgcn_s_memtime();
    __syncthreads();
    if (P) P[7] = __builtin_amdgcn_s_memtime();
    if (wv == 0) {
      if (tid == 0) __hip_atomic_store((GA unsigned*)a.flags + 32 * me, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int rounds = a.wait_all ? (T * T + 63) / 64 : 1;
      for (int r = 0; r < rounds; ++r) {
        int wg = -1;
        if (a.wait_all) { const int q = r * 64 + lane; if (q < T * T) wg = q; }
        else if (lane < T) wg = I * T + lane;
        else if (lane < 2 * T) wg = (lane - T) * T + J;
        unsigned long long t0 = 0;
        for (unsigned spins = 0;; ++spins) {
          const unsigned v = (wg >= 0) ? __hip_atomic_load((GA unsigned*)a.flags + 32 * wg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : target;
          if (__builtin_amdgcn_ballot_w64(v < target) == 0ull) break;
          if (a.sleep > 0) __builtin_amdgcn_s_sleep(1);
          if ((spins & 63u) == 63u) {
            const unsigned long long t = __builtin_amdgcn_s_memrealtime();
            if (t0 == 0) t0 = t;
            else if (t - t0 > kSpinLimit) { if (lane == 0) { abort_s = 1; *a.giveup = 1u; } break; }
          }
        }
      }
    }
    if (P) P[8] = __builtin_amdgcn_s_memtime();
    __syncthreads();
    if (P) P[9] = __builtin_amdgcn_s_memtime();
    return abort_s == 0;
  };

  __syncthreads();
  publish(0, a.tag0);
  bool alive = handover(1u, nullptr);
  const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
  for (int ph = 1; alive && ph <= a.phases; ++ph) {
    unsigned long long P[10];
    P[0] = __builtin_amdgcn_s_memtime();
    const int in = (ph - 1) & 1, out = ph & 1;
    const double want = a.tag0 + (double)((ph - 1) * 1024);
    if (wv < 4) {
      // wave 0: u^T row J (blocks (J, g), written by work-group (g, J))   1: v^T row J   2: u row I (written by (I, g))   3: v row I
      const double* arr = a.buf[in][wv == 0 ? 1 : wv == 1 ? 3 : wv == 2 ? 0 : 2];
      const int row = wv < 2 ? J : I;
      v4d f[T];
#pragma unroll
      for (int g = 0; g < T; ++g) f[g] = LOADS ? ld_frag_sc1(arr, row * T + g, lane) : (v4d){want, want, want, want};
      P[1] = __builtin_amdgcn_s_memtime();
      if (!MFMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      P[2] = __builtin_amdgcn_s_memtime();
      v4d c0[2], c1[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) { c0[h] = (v4d){0.0, 0.0, 0.0, 0.0}; c1[h] = (v4d){0.0, 0.0, 0.0, 0.0}; }
      if (MFMA) {
        const double* p0 = ops + (size_t)((wv >> 1) * 2) * T * 256;
        const double* p1 = p0 + T * 256;
#pragma unroll
        for (int g = 0; g < T; ++g) {
          const v4d k0 = ld_frag_lds(p0 + g * 256, lane), k1 = ld_frag_lds(p1 + g * 256, lane);
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            c0[s & 1] = MFMA_F64(k0[s], f[g][s], c0[s & 1]);
            c1[s & 1] = MFMA_F64(k1[s], f[g][s], c1[s & 1]);
          }
        }
      }
      // every loaded word must carry the phase before this one and the work-group that owns the block
      if (LOADS && a.do_stores) {
#pragma unroll
        for (int g = 0; g < T; ++g) {
          const double w = want + (double)(wv < 2 ? g * T + J : I * T + g);
#pragma unroll
          for (int s = 0; s < 4; ++s) stale += (f[g][s] != w);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        res[((2 * wv) * 4 + r) * 64 + lane] = c0[0][r] + c0[1][r];
        res[((2 * wv + 1) * 4 + r) * 64 + lane] = c1[0][r] + c1[1][r];
      }
    } else {
      P[1] = P[2] = __builtin_amdgcn_s_memtime();
    }
    P[3] = __builtin_amdgcn_s_memtime();
    __syncthreads();
    P[4] = __builtin_amdgcn_s_memtime();
    double tag = a.tag0 + (double)(ph * 1024 + me);
    if (tid < 256) {
      double s = 0.0;
#pragma unroll
      for (int q = 0; q < 8; ++q) s += res[(q * 4 + (wv & 3)) * 64 + lane];
      if (s == 12345.678) tag += 1.0;          // (keeps the reads alive)
    }
    publish(out, tag);
    P[5] = __builtin_amdgcn_s_memtime();
    alive = handover((unsigned)ph + 1u, P);
    if (ph > a.warm) {
#pragma unroll
      for (int k = 0; k < kSeg; ++k) seg[k] += P[k + 1] - P[k];
    }
  }
  const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
  // stale words of the whole work-group
  for (int off = 32; off > 0; off >>= 1) stale += __shfl_xor(stale, off);
  __shared__ unsigned long long stale_s[8];
  if (lane == 0) stale_s[wv] = stale;
  __syncthreads();
  unsigned long long* o = a.out + (size_t)me * kOutPerWg;
  if (lane == 0 && (wv == 0 || wv == 1 || wv == 4)) {
    const int slot = wv == 0 ? 0 : wv == 1 ? 1 : 2;
    for (int k = 0; k < kSeg; ++k) o[slot * kSeg + k] = seg[k];
  }
  if (tid == 0) {
    unsigned long long st = 0;
    for (int w = 0; w < 8; ++w) st += stale_s[w];
    o[3 * kSeg + 0] = st; o[3 * kSeg + 1] = (unsigned long long)abort_s; o[3 * kSeg + 2] = rt1 - rt0;
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Cfg { int T, loads, mfma, stores, st16, wait_all, patched, sleep; const char* tag; };

template <int T, bool LOADS, bool MFMA>
static void launch_t(const Args& a, size_t lds_bytes) {
  static bool attr = false;
  if (!attr) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe<T, LOADS, MFMA>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)); attr = true; }
  hipLaunchKernelGGL((probe<T, LOADS, MFMA>), dim3(T * T), dim3(kThreads), lds_bytes, 0, a);
}
template <int T>
static void launch(const Args& a, size_t lds_bytes) {
  if (a.do_loads && a.do_mfma) launch_t<T, true, true>(a, lds_bytes);
  else if (a.do_loads) launch_t<T, true, false>(a, lds_bytes);
  else if (a.do_mfma) launch_t<T, false, true>(a, lds_bytes);
  else launch_t<T, false, false>(a, lds_bytes);
}

int main(int argc, char** argv) {
  const int phases = argc > 1 ? atoi(argv[1]) : 400;
  const int warm = 20;
  int dev = 0, n_cus = 0;
  CK(hipGetDevice(&dev));
  CK(hipDeviceGetAttribute(&n_cus, hipDeviceAttributeMultiprocessorCount, dev));
  int clk_khz = 0;
  CK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, dev));
  printf("device: %d CUs, clock %d MHz; %d phases (%d warm-up) per configuration; cycles = s_memtime\n", n_cus, clk_khz / 1000, phases, warm);
  const int Tmax = 16;
  Args a;
  memset(&a, 0, sizeof(a));
  const size_t arr_bytes = sizeof(double) * Tmax * Tmax * 256;
  for (int s = 0; s < 2; ++s) for (int q = 0; q < 4; ++q) { CK(hipMalloc(&a.buf[s][q], arr_bytes)); CK(hipMemset(a.buf[s][q], 0, arr_bytes)); }
  CK(hipMalloc(&a.flags, sizeof(unsigned) * 32 * Tmax * Tmax));
  CK(hipMalloc(&a.giveup, 256));
  CK(hipMalloc(&a.out, sizeof(unsigned long long) * kOutPerWg * Tmax * Tmax));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const Cfg cfgs[] = {
      // T loads mfma stores st16 all patched sleep
      {16, 1, 1, 1, 0, 0, 0, 0, "T=16 full skeleton, 8-B stores, mates"},
      {16, 1, 1, 1, 1, 0, 0, 0, "T=16 full skeleton, 16-B stores, mates"},
      {16, 1, 1, 1, 0, 0, 1, 0, "T=16 full skeleton, 8-B stores, mates, XCD-patched tiles"},
      {16, 1, 1, 1, 0, 1, 0, 0, "T=16 full skeleton, 8-B stores, ALL flags"},
      {16, 1, 1, 1, 0, 0, 0, 1, "T=16 full skeleton, 8-B stores, mates, s_sleep in poll"},
      {16, 1, 0, 1, 0, 0, 0, 0, "T=16 no MFMA (loads + stores + flags)"},
      {16, 0, 1, 1, 0, 0, 0, 0, "T=16 no payload loads (MFMA + stores + flags)"},
      {16, 0, 0, 1, 0, 0, 0, 0, "T=16 stores + flags only"},
      {16, 0, 0, 0, 0, 0, 0, 0, "T=16 flags only, mates"},
      {16, 0, 0, 0, 0, 1, 0, 0, "T=16 flags only, ALL"},
      {16, 0, 1, 0, 0, 0, 0, 0, "T=16 MFMA + flags (no payload)"},
      {9, 1, 1, 1, 0, 0, 0, 0, "T=9 full skeleton, 8-B stores, mates"},
      {9, 1, 1, 1, 1, 0, 0, 0, "T=9 full skeleton, 16-B stores, mates"},
      {9, 1, 0, 1, 0, 0, 0, 0, "T=9 no MFMA"},
      {9, 0, 0, 0, 0, 0, 0, 0, "T=9 flags only, mates"},
      {9, 0, 0, 0, 0, 1, 0, 0, "T=9 flags only, ALL"},
  };
  int cfg_no = 0;
  for (const Cfg& c : cfgs) {
    ++cfg_no;
    if (c.T * c.T > n_cus) { printf("%-62s skipped: %d work-groups > %d CUs\n", c.tag, c.T * c.T, n_cus); continue; }
    a.T = c.T; a.phases = phases; a.warm = warm;
    a.do_loads = c.loads; a.do_mfma = c.mfma; a.do_stores = c.stores; a.st16 = c.st16; a.wait_all = c.wait_all;
    a.patched = c.patched; a.sleep = c.sleep;
    a.tag0 = (double)cfg_no * 1048576.0 * 1024.0;
    CK(hipMemset(a.flags, 0, sizeof(unsigned) * 32 * Tmax * Tmax));
    CK(hipMemset(a.giveup, 0, 256));
    CK(hipMemset(a.out, 0, sizeof(unsigned long long) * kOutPerWg * Tmax * Tmax));
    CK(hipDeviceSynchronize());
    const size_t lds_bytes = sizeof(double) * (4 * c.T * 256 + 8 * 256 + 2 * 16 * 17);
    CK(hipEventRecord(e0, 0));
    if (c.T == 16) launch<16>(a, lds_bytes); else launch<9>(a, lds_bytes);
    CK(hipGetLastError());
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const int nwg = c.T * c.T;
    std::vector<unsigned long long> h((size_t)kOutPerWg * nwg);
    CK(hipMemcpy(h.data(), a.out, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
    unsigned long long stale = 0, aborts = 0, rt_max = 0;
    double seg_mean[3][kSeg], seg_max[3][kSeg];
    for (int w = 0; w < 3; ++w) for (int k = 0; k < kSeg; ++k) { seg_mean[w][k] = 0; seg_max[w][k] = 0; }
    const int timed = phases - warm;
    for (int g = 0; g < nwg; ++g) {
      const unsigned long long* o = h.data() + (size_t)g * kOutPerWg;
      stale += o[3 * kSeg]; aborts += o[3 * kSeg + 1]; rt_max = std::max(rt_max, o[3 * kSeg + 2]);
      for (int w = 0; w < 3; ++w) for (int k = 0; k < kSeg; ++k) {
        const double v = (double)o[w * kSeg + k] / timed;
        seg_mean[w][k] += v / nwg; seg_max[w][k] = std::max(seg_max[w][k], v);
      }
    }
    double tot[3] = {0, 0, 0};
    for (int w = 0; w < 3; ++w) for (int k = 0; k < kSeg; ++k) tot[w] += seg_mean[w][k];
    printf("%-62s %4d WGs: %7.3f us/phase (events, incl. launch) %7.3f us/phase (in-kernel)  cycles/phase %7.0f  stale %llu  aborts %llu\n",
           c.tag, nwg, 1e3 * ms / phases, 1e-2 * (double)rt_max / phases, tot[0], stale, aborts);
    static const char* names[kSeg] = {"issue loads", "land (noMFMA)", "MFMA+results", "barrier1", "epilogue+stores", "drain", "barrier2", "flag+poll", "barrier3"};
    static const char* wn[3] = {"wave0", "wave1", "wave4"};
    for (int w = 0; w < 3; ++w) {
      printf("    %s mean[max]:", wn[w]);
      for (int k = 0; k < kSeg; ++k) printf(" %s %.0f[%.0f]", names[k], seg_mean[w][k], seg_max[w][k]);
      printf("\n");
    }
    fflush(stdout);
    if (aborts) { printf("a wait was given up: stopping here\n"); return 2; }
  }
  return 0;
}
