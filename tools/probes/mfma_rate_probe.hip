// mfma_rate_probe.hip -- at what rate does ONE wave issue v_mfma_f64_16x16x4f64, and what changes it?
//
// Why: the chip-wide kernel's contraction waves run 128 MFMAs in ~8 200 cycles (64 per MFMA, the datasheet rate), the
// trial-per-CU kernel's tile waves need ~100 cycles per MFMA -- alone on an idle chip as well as in a batch of 256 -- and
// `ldc_mfma_peak` (eight accumulators, operands in registers, milliseconds long) reads 140 with one wave per SIMD.  The probe
// separates the candidates: number of independent accumulators, an accumulator used twice per k-step (the Laplacians of the
// trial-per-CU kernel), operands from LDS (one ds_read_b64 per operand as there, or none), burst length, waves per SIMD,
// one work-group or the whole chip.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/mfma_rate_probe.hip -o tools/probes/_build/mfma_rate_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef double v4d __attribute__((ext_vector_type(4)));
#define MFMA_F64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// PATTERN 0: CH independent accumulators, one MFMA each per trip, operands in registers
// PATTERN 1: the trip of the trial-per-CU kernel: eight MFMAs on six accumulators (two of them twice), eight operands
// LDSOP: the eight operands of a trip are ds_read_b64 (requested one trip ahead when PIPE), else registers
template <int CH, int PATTERN, bool LDSOP, bool PIPE>
__global__ __launch_bounds__(512) void rate_kernel(double* out, int trips, int bursts, int gap) {
  __shared__ double lds[8 * 64 * 16];
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int q = threadIdx.x; q < 8 * 64 * 16; q += blockDim.x) lds[q] = 1.0 + 1e-9 * q;
  __syncthreads();
  double a = 1.0 + 1e-9 * l, b = 1.0 - 1e-9 * l;
  v4d c[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) c[q] = (v4d){0.0, 0.0, 0.0, (double)q};
  unsigned long long ticks = 0;
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), t00 = __builtin_amdgcn_s_memtime();
  for (int bu = 0; bu < bursts; ++bu) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    double op[8], nx[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) op[q] = LDSOP ? lds[q * 1024 + l] : (q & 1 ? a : b);
    for (int it = 0; it < trips; ++it) {
      if (LDSOP && PIPE) {
#pragma unroll
        for (int q = 0; q < 8; ++q) nx[q] = lds[q * 1024 + ((it + 1) & 15) * 64 + l];
      } else if (LDSOP) {
#pragma unroll
        for (int q = 0; q < 8; ++q) op[q] = lds[q * 1024 + (it & 15) * 64 + l];
      }
      if (PATTERN == 0) {
#pragma unroll
        for (int q = 0; q < CH; ++q) c[q] = MFMA_F64(op[q & 7], op[(q + 1) & 7], c[q]);
      } else {
        c[0] = MFMA_F64(op[0], op[2], c[0]);
        c[1] = MFMA_F64(op[1], op[2], c[1]);
        c[2] = MFMA_F64(op[0], op[3], c[2]);
        c[3] = MFMA_F64(op[1], op[3], c[3]);
        c[4] = MFMA_F64(op[4], op[6], c[4]);
        c[1] = MFMA_F64(op[4], op[7], c[1]);
        c[5] = MFMA_F64(op[5], op[6], c[5]);
        c[3] = MFMA_F64(op[5], op[7], c[3]);
      }
      if (LDSOP && PIPE) {
#pragma unroll
        for (int q = 0; q < 8; ++q) op[q] = nx[q];
      }
    }
    // (the results must exist before the stamp: one dependent instruction per accumulator)
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) s += c[q][0];
    asm volatile("" :: "v"(s));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    ticks += t1 - t0;
    for (int g = 0; g < gap; ++g) __builtin_amdgcn_s_sleep(16);      // ~1 000 cycles per unit
  }
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime(), t11 = __builtin_amdgcn_s_memtime();
  double s = 0.0;
#pragma unroll
  for (int q = 0; q < 8; ++q) s += c[q][0] + c[q][1] + c[q][2] + c[q][3];
  if (s == 123.456) out[0] = s;
  if (l == 0) {
    double* o = out + 8 + ((size_t)blockIdx.x * 8 + w) * 4;
    o[0] = (double)ticks; o[1] = (double)(r1 - r0); o[2] = (double)(t11 - t00);
  }
}

template <int CH, int PATTERN, bool LDSOP, bool PIPE>
void run(const char* label, double* d_out, int grid, int waves, int trips, int bursts, int gap) {
  const int per_trip = PATTERN == 0 ? CH : 8;
  std::vector<double> h(8 + (size_t)grid * 8 * 4);
  CHECK(hipMemset(d_out, 0, h.size() * sizeof(double)));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((rate_kernel<CH, PATTERN, LDSOP, PIPE>), dim3(grid), dim3(64 * waves), 0, 0, d_out, trips, 2, 0);   // warm
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((rate_kernel<CH, PATTERN, LDSOP, PIPE>), dim3(grid), dim3(64 * waves), 0, 0, d_out, trips, bursts, gap);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  CHECK(hipMemcpy(h.data(), d_out, h.size() * sizeof(double), hipMemcpyDeviceToHost));
  double tmin = 1e30, tmax = 0, tsum = 0, clk = 0;
  int n = 0;
  for (int g = 0; g < grid; ++g)
    for (int w = 0; w < waves; ++w) {
      const double* o = h.data() + 8 + ((size_t)g * 8 + w) * 4;
      const double per = o[0] / ((double)bursts * trips * per_trip);
      tmin = per < tmin ? per : tmin; tmax = per > tmax ? per : tmax; tsum += per; ++n;
      clk += o[2] / (o[1] * 10e-9) / 1e6;
    }
  const double mfmas = (double)grid * waves * bursts * trips * per_trip;
  printf("%-58s grid %3d waves/SIMD %d burst %6d MFMAs x %4d gap %3d: %6.1f cycles/MFMA/wave (min %6.1f max %6.1f) = %6.1f per SIMD, clock %4.0f MHz, %6.2f TFLOP/s over the launch\n",
         label, grid, waves / 4, trips * per_trip, bursts, gap, tsum / n, tmin, tmax, tsum / n / (waves / 4), clk / n,
         mfmas * 2048.0 / (ms * 1e-3) / 1e12);
  fflush(stdout);
}

int main() {
  double* d_out;
  CHECK(hipMalloc(&d_out, sizeof(double) * (8 + 256 * 8 * 4)));
  for (int grid : {1, 256}) {
    for (int waves : {4, 8}) {
      // short bursts (the length of a chip-wide kernel's stage), long bursts (a trial-per-CU stage), one long run
      run<4, 0, false, false>("4 accumulators, operands in registers", d_out, grid, waves, 32, 64, 0);
      run<4, 0, false, false>("4 accumulators, operands in registers", d_out, grid, waves, 32, 64, 8);
      run<8, 0, false, false>("8 accumulators, operands in registers", d_out, grid, waves, 16, 64, 0);
      run<8, 0, false, false>("8 accumulators, operands in registers", d_out, grid, waves, 16, 64, 8);
      run<8, 0, false, false>("8 accumulators, operands in registers", d_out, grid, waves, 20000, 1, 0);
      run<8, 1, false, false>("6 accumulators (two twice), operands in registers", d_out, grid, waves, 11, 64, 4);
      run<8, 1, true, false>("6 accumulators (two twice), 8 ds_read_b64 per trip", d_out, grid, waves, 11, 64, 4);
      run<8, 1, true, true>("6 accumulators (two twice), 8 ds_read_b64 a trip ahead", d_out, grid, waves, 11, 64, 4);
      run<8, 0, true, true>("8 accumulators, 8 ds_read_b64 a trip ahead", d_out, grid, waves, 11, 64, 4);
      run<4, 0, true, true>("4 accumulators, 8 ds_read_b64 a trip ahead", d_out, grid, waves, 22, 64, 4);
    }
  }
  return 0;
}
