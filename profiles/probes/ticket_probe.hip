// ticket_probe.hip -- what does "the last workgroup forms the ordered block sum" cost against a second launch?
// G workgroups of 256 threads each write one partial (the shape of the solver kernels' stage-1 reduction).  Variants:
//   sep     : partial kernel, then a separate one-workgroup ordered-sum kernel (round-1 form: k_block_sums)
//   ticket1 : __threadfence + one agent-scope atomic counter; the workgroup that draws the last ticket sums
//   ticket2 : two levels -- one counter per 64 workgroups (own 64-byte line), the group's last arrival bumps a top counter
// Build/run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o ticket_probe ticket_probe.hip && ./ticket_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define T 256
__device__ __forceinline__ double ordered_sum(const double *p, int n) {
  __shared__ double sh[T];
  const int t = threadIdx.x;
  double v = 0.0;
  for (int c = t; c < n; c += T) v = v + p[c];
  sh[t] = v;
  __syncthreads();
  for (int s = T / 2; s > 0; s >>= 1) { if (t < s) sh[t] = sh[t] + sh[t + s]; __syncthreads(); }
  return sh[0];
}
__device__ __forceinline__ double wg_partial(const double *x, size_t n) {
  __shared__ double sh[T];
  const int t = threadIdx.x;
  const size_t q = (size_t)blockIdx.x * T + t;
  sh[t] = q < n ? x[q] * 0.5 : 0.0;
  __syncthreads();
  for (int s = T / 2; s > 0; s >>= 1) { if (t < s) sh[t] = sh[t] + sh[t + s]; __syncthreads(); }
  return sh[0];
}
__global__ void __launch_bounds__(T) k_partial(const double *x, size_t n, double *partial) {
  const double p = wg_partial(x, n);
  if (threadIdx.x == 0) partial[blockIdx.x] = p;
}
__global__ void __launch_bounds__(T) k_sum(const double *partial, int n, double *out) {
  const double s = ordered_sum(partial, n);
  if (threadIdx.x == 0) *out = s;
}
__global__ void __launch_bounds__(T) k_ticket1(const double *x, size_t n, double *partial, unsigned *ctr, double *out) {
  __shared__ int last;
  const double p = wg_partial(x, n);
  if (threadIdx.x == 0) {
    partial[blockIdx.x] = p;
    __threadfence();
    last = atomicAdd(ctr, 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (last) {
    __threadfence();
    const double s = ordered_sum(partial, gridDim.x);
    if (threadIdx.x == 0) { *out = s; *ctr = 0; }
  }
}
__global__ void __launch_bounds__(T) k_ticket2(const double *x, size_t n, double *partial, unsigned *ctr, double *out) {
  __shared__ int last;
  const double p = wg_partial(x, n);
  if (threadIdx.x == 0) {
    partial[blockIdx.x] = p;
    __threadfence();
    const unsigned grp = blockIdx.x >> 6, ngrp = (gridDim.x + 63) >> 6;
    const unsigned gsize = grp == ngrp - 1 ? gridDim.x - (grp << 6) : 64u;
    int l = 0;
    if (atomicAdd(ctr + 16 * (1 + grp), 1u) == gsize - 1) {
      ctr[16 * (1 + grp)] = 0;
      __threadfence();
      l = atomicAdd(ctr, 1u) == ngrp - 1;
    }
    last = l;
  }
  __syncthreads();
  if (last) {
    __threadfence();
    const double s = ordered_sum(partial, gridDim.x);
    if (threadIdx.x == 0) { *out = s; *ctr = 0; }
  }
}
// round 3: no agent-scope fence at all -- the partial goes out as an agent-scope (write-through, sc1) relaxed atomic store, a
// workgroup-scope release only waits for it, the ticket is a relaxed agent-scope atomic, the summing workgroup reads the partials
// with agent-scope relaxed atomic loads
__device__ __forceinline__ double ordered_sum_sc1(const double *p, int n) {
  __shared__ double sh[T];
  const int t = threadIdx.x;
  double v = 0.0;
  for (int c = t; c < n; c += T) v = v + __hip_atomic_load(p + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  sh[t] = v;
  __syncthreads();
  for (int s = T / 2; s > 0; s >>= 1) { if (t < s) sh[t] = sh[t] + sh[t + s]; __syncthreads(); }
  return sh[0];
}
__global__ void __launch_bounds__(T) k_ticket3(const double *x, size_t n, double *partial, unsigned *ctr, double *out) {
  __shared__ int last;
  const double p = wg_partial(x, n);
  if (threadIdx.x == 0) {
    __hip_atomic_store(partial + blockIdx.x, p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    last = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
  }
  __syncthreads();
  if (last) {
    const double s = ordered_sum_sc1(partial, gridDim.x);
    if (threadIdx.x == 0) { *out = s; __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  }
}
__global__ void __launch_bounds__(T) k_ticket4(const double *x, size_t n, double *partial, unsigned *ctr, double *out) {
  __shared__ int last;
  const double p = wg_partial(x, n);
  if (threadIdx.x == 0) {
    __hip_atomic_store(partial + blockIdx.x, p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    const unsigned grp = blockIdx.x >> 6, ngrp = (gridDim.x + 63) >> 6;
    const unsigned gsize = grp == ngrp - 1 ? gridDim.x - (grp << 6) : 64u;
    int l = 0;
    if (__hip_atomic_fetch_add(ctr + 16 * (1 + grp), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gsize - 1) {
      __hip_atomic_store(ctr + 16 * (1 + grp), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      l = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ngrp - 1;
    }
    last = l;
  }
  __syncthreads();
  if (last) {
    const double s = ordered_sum_sc1(partial, gridDim.x);
    if (threadIdx.x == 0) { *out = s; __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  }
}
int main() {
  for (int G : {480, 4224, 33792}) {
    const size_t n = (size_t)G * T;
    double *x, *partial, *out; unsigned *ctr;
    hipMalloc(&x, n * 8); hipMalloc(&partial, G * 8); hipMalloc(&out, 64); hipMalloc(&ctr, 4 * 16 * (2 + G / 64));
    hipMemset(ctr, 0, 4 * 16 * (2 + G / 64));
    std::vector<double> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = 1.0 + (i % 7) * 0.125;
    hipMemcpy(x, h.data(), n * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 200;
    double res[6] = {}; float ms[6] = {};
    for (int v = 0; v < 6; ++v) {
      for (int r = -20; r < reps; ++r) {
        if (r == 0) hipEventRecord(e0, 0);
        if (v == 0) hipLaunchKernelGGL(k_partial, dim3(G), dim3(T), 0, 0, x, n, partial);
        if (v == 1) { hipLaunchKernelGGL(k_partial, dim3(G), dim3(T), 0, 0, x, n, partial); hipLaunchKernelGGL(k_sum, dim3(1), dim3(T), 0, 0, partial, G, out); }
        if (v == 2) hipLaunchKernelGGL(k_ticket1, dim3(G), dim3(T), 0, 0, x, n, partial, ctr, out);
        if (v == 3) hipLaunchKernelGGL(k_ticket2, dim3(G), dim3(T), 0, 0, x, n, partial, ctr, out);
        if (v == 4) hipLaunchKernelGGL(k_ticket3, dim3(G), dim3(T), 0, 0, x, n, partial, ctr, out);
        if (v == 5) hipLaunchKernelGGL(k_ticket4, dim3(G), dim3(T), 0, 0, x, n, partial, ctr, out);
      }
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms[v], e0, e1);
      hipMemcpy(&res[v], out, 8, hipMemcpyDeviceToHost);
    }
    printf("G=%6d  partial only %.2f us | sep (2 launches) %.2f us | ticket1 %.2f us | ticket2 %.2f us | sc1 ticket 1-level %.2f us | sc1 ticket 2-level %.2f us   sums %.6f %.6f %.6f %.6f %.6f\n", G,
           1e3 * ms[0] / reps, 1e3 * ms[1] / reps, 1e3 * ms[2] / reps, 1e3 * ms[3] / reps, 1e3 * ms[4] / reps, 1e3 * ms[5] / reps, res[1], res[2], res[3], res[4], res[5]);
    hipFree(x); hipFree(partial); hipFree(out); hipFree(ctr);
  }
  return 0;
}
