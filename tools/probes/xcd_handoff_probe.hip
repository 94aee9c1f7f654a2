// xcd_handoff_probe.hip -- what does a flag hand-off between two work-groups of ONE XCD cost, by store / load flavour?
//
// The small-N trial kernel (csrc/ldc_xcd_kernel.inc) spends ~2 400-3 300 cycles per stage between "my stores are drained"
// and "I have seen my mates' flags".  This probe plays ping-pong between two single-wave work-groups that found themselves
// on the same XCD (HW_REG_XCC_ID, one pair per XCD, all eight pairs at once) and reports cycles per ROUND TRIP (two
// hand-offs) for every combination of
//   store: 0 plain   1 sc1 (write-through)   2 L2 atomic add (no return)
//   load : 0 sc1     1 nt                    2 sc0 sc1      3 returning L2 atomic OR     4 plain after buffer_inv sc1
// and, for the payload case, whether a 2-KB block written with plain stores before the flag is read back FRESH by
// sc1 / nt loads after the flag has been seen (mismatches counted).
//   hipcc --offload-arch=gfx950 -O2 tools/probes/xcd_handoff_probe.hip -o tools/probes/_build/xcd_handoff_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define GA __attribute__((address_space(1)))
constexpr int kMaxPolls = 200000;

__device__ __forceinline__ void st_flag(unsigned* p, unsigned v, int flavour) {
  if (flavour == 0) asm volatile("global_store_dword %0, %1, off" ::"v"((GA unsigned*)p), "v"(v) : "memory");
  else if (flavour == 1) asm volatile("global_store_dword %0, %1, off sc1" ::"v"((GA unsigned*)p), "v"(v) : "memory");
  else { const unsigned one = 1u; asm volatile("global_atomic_add %0, %1, off" ::"v"((GA unsigned*)p), "v"(one) : "memory"); }
}
__device__ __forceinline__ unsigned ld_flag(unsigned* p, int flavour) {
  unsigned r;
  const unsigned zero = 0u;
  switch (flavour) {
    case 0: asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"((GA unsigned*)p) : "memory"); break;
    case 1: asm volatile("global_load_dword %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"((GA unsigned*)p) : "memory"); break;
    case 2: asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"((GA unsigned*)p) : "memory"); break;
    case 3: asm volatile("global_atomic_or %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(r) : "v"((GA unsigned*)p), "v"(zero) : "memory"); break;
    default: asm volatile("buffer_inv sc1\n\tglobal_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"((GA unsigned*)p) : "memory"); break;
  }
  return r;
}
__device__ __forceinline__ bool wait_flag(unsigned* p, unsigned want, int flavour, int sleep) {
  for (int k = 0; k < kMaxPolls; ++k) {
    if (ld_flag(p, flavour) >= want) return true;
    if (sleep == 1) __builtin_amdgcn_s_sleep(1); else if (sleep == 4) __builtin_amdgcn_s_sleep(4); else if (sleep == 16) __builtin_amdgcn_s_sleep(16);
  }
  return false;
}

// ws: [0..511] tickets (one line per XCD), [512 + 64 x .. ] ping flag line, +32 pong flag line; payload at ws + 4096 + 1024 x (uint32)
__global__ __launch_bounds__(64) void probe(unsigned* ws, int wf, int rf, int sleep, int rounds, int payload_load,
                                            unsigned long long* out) {
  extern __shared__ double pad[];
  const int lane = threadIdx.x;
  __shared__ int role_s, xcc_s;
  if (lane == 0) {
    const unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 15u;
    role_s = (int)atomicAdd(ws + 32 * xcc, 1u);
    xcc_s = (int)xcc;
  }
  __syncthreads();
  const int role = role_s, xcc = xcc_s;
  if (role > 1) return;
  unsigned* ping = ws + 512 + 64 * xcc;
  unsigned* pong = ping + 32;
  unsigned* pay = ws + 4096 + 1024 * xcc;           // 2 KB + a second 2 KB for the way back
  unsigned long long bad = 0, fail = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 1; r <= rounds; ++r) {
    unsigned* mine = role == 0 ? ping : pong;
    unsigned* theirs = role == 0 ? pong : ping;
    unsigned* wpay = pay + (role == 0 ? 0 : 512);
    unsigned* rpay = pay + (role == 0 ? 512 : 0);
    if (role == 1) {          // pong waits first
      if (lane == 0 && !wait_flag(theirs, (unsigned)r, rf, sleep)) fail = 1;
      fail = __builtin_amdgcn_readfirstlane((int)fail);
      if (fail) break;
      if (payload_load >= 0) {
        unsigned v[8];
        for (int q = 0; q < 8; ++q) {
          if (payload_load == 0) asm volatile("global_load_dword %0, %1, off sc1" : "=v"(v[q]) : "v"((GA unsigned*)(rpay + lane * 8 + q)) : "memory");
          else if (payload_load == 1) asm volatile("global_load_dword %0, %1, off nt" : "=v"(v[q]) : "v"((GA unsigned*)(rpay + lane * 8 + q)) : "memory");
          else asm volatile("global_load_dword %0, %1, off" : "=v"(v[q]) : "v"((GA unsigned*)(rpay + lane * 8 + q)) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int q = 0; q < 8; ++q) bad += (v[q] != (unsigned)(r * 1000 + lane * 8 + q));
      }
    }
    if (payload_load >= 0) {
      for (int q = 0; q < 8; ++q) wpay[lane * 8 + q] = (unsigned)(r * 1000 + lane * 8 + q);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (lane == 0) st_flag(mine, (unsigned)r, wf);
    if (role == 0) {
      if (lane == 0 && !wait_flag(theirs, (unsigned)r, rf, sleep)) fail = 1;
      fail = __builtin_amdgcn_readfirstlane((int)fail);
      if (fail) break;
      if (payload_load >= 0) {
        unsigned v[8];
        for (int q = 0; q < 8; ++q) {
          if (payload_load == 0) asm volatile("global_load_dword %0, %1, off sc1" : "=v"(v[q]) : "v"((GA unsigned*)(rpay + lane * 8 + q)) : "memory");
          else if (payload_load == 1) asm volatile("global_load_dword %0, %1, off nt" : "=v"(v[q]) : "v"((GA unsigned*)(rpay + lane * 8 + q)) : "memory");
          else asm volatile("global_load_dword %0, %1, off" : "=v"(v[q]) : "v"((GA unsigned*)(rpay + lane * 8 + q)) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int q = 0; q < 8; ++q) bad += (v[q] != (unsigned)(r * 1000 + lane * 8 + q));
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  for (int off = 32; off > 0; off >>= 1) bad += __shfl_xor(bad, off);
  if (role == 0 && lane == 0) {
    out[3 * xcc + 0] = fail ? ~0ull : (t1 - t0) / (unsigned long long)rounds;
    out[3 * xcc + 1] = bad;
    out[3 * xcc + 2] = 1;
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 2000;
  unsigned* ws; unsigned long long* out;
  const size_t wsb = sizeof(unsigned) * (4096 + 1024 * 16);
  CK(hipMalloc(&ws, wsb)); CK(hipMalloc(&out, sizeof(unsigned long long) * 48));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
  const char* wn[3] = {"plain store", "sc1 store", "L2 atomic add"};
  const char* rn[5] = {"sc1 load", "nt load", "sc0 sc1 load", "returning L2 atomic", "buffer_inv sc1 + plain load"};
  auto run = [&](int wf, int rf, int sleep, int pl, const char* tag) {
    CK(hipMemset(ws, 0, wsb)); CK(hipMemset(out, 0, sizeof(unsigned long long) * 48));
    hipLaunchKernelGGL(probe, dim3(8 * 6), dim3(64), 100 * 1024, 0, ws, wf, rf, sleep, rounds, pl, out);
    CK(hipDeviceSynchronize());
    unsigned long long h[48];
    CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
    std::vector<unsigned long long> v; unsigned long long bad = 0; int timeouts = 0;
    for (int x = 0; x < 16; ++x) if (h[3 * x + 2]) { if (h[3 * x] == ~0ull) ++timeouts; else v.push_back(h[3 * x]); bad += h[3 * x + 1]; }
    unsigned long long mn = ~0ull, mx = 0, sum = 0;
    for (auto c : v) { mn = c < mn ? c : mn; mx = c > mx ? c : mx; sum += c; }
    printf("%-34s %-13s + %-27s sleep %2d: pairs %zu  cycles/round trip mean %6llu [min %6llu max %6llu]  timeouts %d  stale words %llu\n",
           tag, wn[wf], rn[rf], sleep, v.size(), v.empty() ? 0ull : sum / v.size(), v.empty() ? 0ull : mn, mx, timeouts, bad);
    fflush(stdout);
  };
  for (int wf = 0; wf < 3; ++wf)
    for (int rf = 0; rf < 5; ++rf) run(wf, rf, 0, -1, "flag only");
  for (int sleep : {1, 4, 16}) { run(0, 0, sleep, -1, "flag only"); run(2, 3, sleep, -1, "flag only"); }
  for (int pl = 0; pl < 3; ++pl) {
    static const char* tags[3] = {"2 KB payload read back by sc1", "2 KB payload read back by nt", "2 KB payload read back plain"};
    run(0, 0, 0, pl, tags[pl]);
    run(0, 1, 0, pl, tags[pl]);
    run(2, 3, 0, pl, tags[pl]);
  }
  return 0;
}
