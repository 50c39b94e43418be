// Probe for kernels_pcg_persist.hpp: the 16-byte {value, tag} words.  (1) round trip of st_pword / ld_pwords14 across a launch boundary;
// (2) torn-access check: writer workgroups keep rewriting words with (value = f(tag), tag) while reader workgroups poll them -- every word read
// must be self-consistent.   hipcc --offload-arch=gfx950 -O3 -I pop2-cesm_amd/csrc -I include profiles/probes/pword_probe.hip -o /tmp/pword_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "kernels_pcg_persist.hpp"
using namespace pop;
__global__ void k_write(PWord *W, int n, unsigned long long tag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) st_pword(W + i, 1.0 + i * 0.5, tag + i);
}
__global__ void k_read(const PWord *W, int n, double *val, unsigned long long *tag) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned off[14]; pword4 got[14];
  for (int k = 0; k < 14; ++k) off[k] = (unsigned)(((t * 14 + k) % n) * sizeof(PWord));
  ld_pwords14(W, off, got);
  for (int k = 0; k < 14; ++k) { val[t * 14 + k] = pword_value(got[k]); tag[t * 14 + k] = pword_tag(got[k]); }
}
// torn-access check: blocks 0..NWR-1 write, the others read
__global__ void k_torn(PWord *W, int n, int nwr, int rounds, unsigned long long *bad) {
  const int t = threadIdx.x;
  if ((int)blockIdx.x < nwr) {
    for (int r = 1; r <= rounds; ++r)
      for (int i = blockIdx.x * blockDim.x + t; i < n; i += nwr * blockDim.x) st_pword(W + i, (double)r * 3.0 + i, (unsigned long long)r);
  } else {
    unsigned off[14]; pword4 got[14];
    unsigned long long nb = 0;
    for (int r = 0; r < rounds; ++r) {
      for (int k = 0; k < 14; ++k) off[k] = (unsigned)(((t * 14 + k + r * 7 + blockIdx.x * 131) % n) * sizeof(PWord));
      ld_pwords14(W, off, got);
      for (int k = 0; k < 14; ++k) {
        const int i = off[k] / sizeof(PWord);
        const unsigned long long g = pword_tag(got[k]);
        const double v = pword_value(got[k]);
        if (g != 0 && v != (double)g * 3.0 + i) ++nb;
      }
    }
    if (nb) atomicAdd(bad, nb);
  }
}
int main() {
  const int n = 4096;
  PWord *W; double *val; unsigned long long *tag, *bad;
  hipMalloc(&W, n * sizeof(PWord)); hipMemset(W, 0, n * sizeof(PWord));
  hipMalloc(&val, 256 * 14 * 8); hipMalloc(&tag, 256 * 14 * 8); hipMalloc(&bad, 8); hipMemset(bad, 0, 8);
  hipLaunchKernelGGL(k_write, dim3(n / 256), dim3(256), 0, 0, W, n, 1000ULL << 32);
  hipLaunchKernelGGL(k_read, dim3(1), dim3(256), 0, 0, W, n, val, tag);
  std::vector<double> hv(256 * 14); std::vector<unsigned long long> ht(256 * 14);
  hipMemcpy(hv.data(), val, hv.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(ht.data(), tag, ht.size() * 8, hipMemcpyDeviceToHost);
  int wrong = 0;
  for (int t = 0; t < 256; ++t) for (int k = 0; k < 14; ++k) {
    const int i = (t * 14 + k) % n;
    if (hv[t * 14 + k] != 1.0 + i * 0.5 || ht[t * 14 + k] != (1000ULL << 32) + i) { if (wrong < 5) printf("t %d k %d i %d: %g %llx\n", t, k, i, hv[t * 14 + k], ht[t * 14 + k]); ++wrong; }
  }
  printf("round trip: %d wrong of %d\n", wrong, 256 * 14);
  hipMemset(W, 0, n * sizeof(PWord));
  hipLaunchKernelGGL(k_torn, dim3(64), dim3(256), 0, 0, W, n, 16, 2000, bad);
  unsigned long long hb = 0; hipDeviceSynchronize(); hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost);
  printf("torn words seen: %llu\n", hb);
  return (wrong || hb) ? 1 : 0;
}
