#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v2d __attribute__((ext_vector_type(2)));
__global__ void probe(const double* src, double* out) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double* mine = lds + wv * 512;   // 4 KB per wave
  // each lane: 16 bytes from src + lane*4 doubles (+0 and +2 doubles)
  const double* g = src + wv * 1000 + lane * 4;
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)mine, 16, 0, 0);
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + 2),
                                   (__attribute__((address_space(3))) void*)(mine + 128), 16, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);   // everything
  __syncthreads();
  v2d a = *reinterpret_cast<v2d*>(mine + 2 * lane);
  v2d b = *reinterpret_cast<v2d*>(mine + 128 + 2 * lane);
  out[(wv * 64 + lane) * 4 + 0] = a[0]; out[(wv * 64 + lane) * 4 + 1] = a[1];
  out[(wv * 64 + lane) * 4 + 2] = b[0]; out[(wv * 64 + lane) * 4 + 3] = b[1];
}
int main() {
  const int n = 8 * 1000 + 1024;
  std::vector<double> h(n); for (int i = 0; i < n; ++i) h[i] = i;
  double *d, *o; (void)hipMalloc(&d, n * 8); (void)hipMalloc(&o, 512 * 4 * 8);
  (void)hipMemcpy(d, h.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(512), 8 * 4096, 0, d, o);
  std::vector<double> r(2048); (void)hipMemcpy(r.data(), o, 2048 * 8, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int w = 0; w < 8; ++w) for (int l = 0; l < 64; ++l) for (int k = 0; k < 4; ++k) {
    double e = w * 1000 + l * 4 + k; if (r[(w * 64 + l) * 4 + k] != e) { if (bad < 8) printf("w%d l%d k%d got %g want %g\n", w, l, k, r[(w*64+l)*4+k], e); ++bad; }
  }
  printf("dma probe: %d mismatches\n", bad);
  return bad != 0;
}
