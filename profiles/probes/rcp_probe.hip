// rcp_probe.hip -- is 1.0 / x (the 11-instruction IEEE division sequence) reproduced bit for bit by v_rcp_f64 followed by
// Newton steps with explicit fma, for the arguments the MWJF equation of state divides by (W2 in [0.5, 2])?
//   nr1 : r = rcp(x); e = fma(-x, r, 1); r = fma(r, e, r)
//   nr2 : nr1 + a second step (e = fma(-x, r, 1); r = fma(r, e, r))
//   nr3 : three steps
// Counts the values whose result differs from 1.0 / x, over N random and N "hard" (significand near all-ones) arguments,
// and times each variant (dependent chains of 64 reciprocals per thread).
// Build/run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o rcp_probe rcp_probe.hip && ./rcp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
template <int NR> __device__ __forceinline__ double rcp_nr(double x) {
  double r = __builtin_amdgcn_rcp(x);
#pragma unroll
  for (int t = 0; t < NR; ++t) { const double e = __builtin_fma(-x, r, 1.0); r = __builtin_fma(r, e, r); }
  return r;
}
template <int NR> __global__ void k_cmp(const double *x, size_t n, unsigned long long *bad, double *worst) {
  const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  const double a = 1.0 / x[q], b = (NR == 0) ? __builtin_amdgcn_rcp(x[q]) : rcp_nr<NR>(x[q]);
  if (a != b) {
    atomicAdd(bad, 1ULL);
    const double rel = fabs(a - b) / fabs(a);
    // racy max is fine for a report
    if (rel > *worst) *worst = rel;
  }
}
template <int NR> __global__ void k_time(const double *x, size_t n, double *out) {
  const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  double v = x[q], s = 0.0;
#pragma unroll 1
  for (int t = 0; t < 64; ++t) {
    const double r = (NR < 0) ? 1.0 / v : rcp_nr<(NR < 0 ? 1 : NR)>(v);
    s = s + r;
    v = 0.75 + 0.5 * r;     // stays in [1, 1.75]
  }
  out[q] = s;
}
int main() {
  const size_t N = 1u << 26;
  std::vector<double> h(2 * N);
  uint64_t st = 0x9E3779B97F4A7C15ULL;
  auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
  for (size_t i = 0; i < N; ++i) {            // uniform significands, exponents 2^-1 .. 2^0
    const uint64_t m = rnd() & ((1ULL << 52) - 1), e = 1022 + (rnd() & 1);
    const uint64_t bits = (e << 52) | m;
    memcpy(&h[i], &bits, 8);
  }
  for (size_t i = 0; i < N; ++i) {            // hard cases: top 30 significand bits all ones / all zeros, random tail
    const uint64_t tail = rnd() & ((1ULL << 22) - 1);
    const uint64_t m = (i & 1) ? (((1ULL << 30) - 1) << 22) | tail : tail;
    const uint64_t bits = (1023ULL << 52) | m;
    memcpy(&h[N + i], &bits, 8);
  }
  double *x, *out, *worst; unsigned long long *bad;
  hipMalloc(&x, 2 * N * 8); hipMalloc(&out, 2 * N * 8); hipMalloc(&bad, 8); hipMalloc(&worst, 8);
  hipMemcpy(x, h.data(), 2 * N * 8, hipMemcpyHostToDevice);
  const int T = 256; const unsigned G = (unsigned)((2 * N + T - 1) / T);
  auto cmp = [&](auto kern, const char *name) {
    hipMemset(bad, 0, 8); hipMemset(worst, 0, 8);
    hipLaunchKernelGGL(kern, dim3(G), dim3(T), 0, 0, x, 2 * N, bad, worst);
    unsigned long long b; double w;
    hipMemcpy(&b, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&w, worst, 8, hipMemcpyDeviceToHost);
    printf("%-6s differs from 1.0/x in %llu of %zu values (worst rel %.3e)\n", name, b, 2 * N, w);
  };
  cmp(k_cmp<0>, "rcp"); cmp(k_cmp<1>, "nr1"); cmp(k_cmp<2>, "nr2"); cmp(k_cmp<3>, "nr3");
  auto tim = [&](auto kern, const char *name) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(kern, dim3(G), dim3(T), 0, 0, x, 2 * N, out);
    hipEventRecord(a, 0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(G), dim3(T), 0, 0, x, 2 * N, out);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-6s %.3f ms per launch (64 dependent reciprocals x %zu threads)\n", name, ms / 5, 2 * N);
  };
  tim(k_time<-1>, "div"); tim(k_time<1>, "nr1"); tim(k_time<2>, "nr2"); tim(k_time<3>, "nr3");
  return 0;
}
