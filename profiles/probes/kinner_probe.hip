// kinner_probe.hip -- the layout north_star names, measured: (k, i, j) storage with k innermost and ONE WAVEFRONT PER
// WATER COLUMN (lane = level; 62 of 64 lanes active, a column is 496 contiguous bytes that start 16 B x (column mod 8) off
// a 128-byte line), against the layout this build uses: (i, j, k) with i fastest, one thread per column, a wave = 64
// consecutive i of one level (512 aligned bytes).  Both kernels stream the same 10 input + 2 output fp64 fields of
// tx0.1v3 size and do the cross-level work of a scan the cheapest possible way (k-innermost: six cross-lane shuffle
// steps per array pair; i-fastest: a running sum in a register), so what is compared is the memory system.
//   hipcc --offload-arch=gfx950 -O3 -o kinner_probe kinner_probe.hip && ./kinner_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define NA 10
struct Arrs { const double *a[NA]; double *o[2]; };
// k innermost, one wave per column, `cols_per_wave` columns per wave in a grid-stride loop
__global__ void __launch_bounds__(256) kinner(Arrs A, size_t ncol, int km) {
  const int lane = threadIdx.x & 63;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwave = ((size_t)gridDim.x * blockDim.x) >> 6;
  for (size_t c = wave; c < ncol; c += nwave) {
    double s0 = 0, s1 = 0;
    if (lane < km) {
#pragma unroll
      for (int a = 0; a < NA; a += 2) { s0 += A.a[a][c * km + lane]; s1 += A.a[a + 1][c * km + lane]; }
    }
    // inclusive scan over the levels (what a one-wave-per-column Thomas / hydrostatic sum needs): 6 shuffle steps
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const double t0 = __shfl_up(s0, d), t1 = __shfl_up(s1, d);
      if (lane >= d) { s0 += t0; s1 += t1; }
    }
    if (lane < km) { A.o[0][c * km + lane] = s0; A.o[1][c * km + lane] = s1; }
  }
}
// i fastest, one thread per column, marching k with a running sum (the shape of this build's column kernels)
__global__ void __launch_bounds__(64) ifast(Arrs A, size_t n2, int km) {
  const size_t p2 = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (p2 >= n2) return;
  double r0 = 0, r1 = 0, nxt[NA];
#pragma unroll
  for (int a = 0; a < NA; ++a) nxt[a] = A.a[a][p2];
  for (int k = 0; k < km; ++k) {
    double cur[NA];
#pragma unroll
    for (int a = 0; a < NA; ++a) cur[a] = nxt[a];
    const int kn = k + 1 < km ? k + 1 : k;
#pragma unroll
    for (int a = 0; a < NA; ++a) nxt[a] = A.a[a][(size_t)kn * n2 + p2];
#pragma unroll
    for (int a = 0; a < NA; a += 2) { r0 += cur[a]; r1 += cur[a + 1]; }
    A.o[0][(size_t)k * n2 + p2] = r0; A.o[1][(size_t)k * n2 + p2] = r1;
  }
}
int main() {
  const int nx = 3604, ny = 2404, km = 62;
  const size_t n2 = (size_t)nx * ny, n = n2 * km;
  Arrs A;
  for (int a = 0; a < NA; ++a) { double *p; if (hipMalloc(&p, n * 8) != hipSuccess) { printf("alloc failed\n"); return 1; } hipMemset(p, 0, n * 8); A.a[a] = p; }
  for (int a = 0; a < 2; ++a) { double *p; hipMalloc(&p, n * 8); A.o[a] = p; }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double gb = 12.0 * n * 8 / 1e9;
  for (int v = 0; v < 4; ++v) {
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
      hipEventRecord(e0);
      if (v == 0) hipLaunchKernelGGL(ifast, dim3((unsigned)((n2 + 63) / 64)), dim3(64), 0, 0, A, n2, km);
      if (v == 1) hipLaunchKernelGGL(kinner, dim3(256 * 8), dim3(256), 0, 0, A, n2, km);       // 8 workgroups per CU, grid-stride
      if (v == 2) hipLaunchKernelGGL(kinner, dim3(256 * 32), dim3(256), 0, 0, A, n2, km);      // more waves in flight
      if (v == 3) hipLaunchKernelGGL(kinner, dim3((unsigned)((n2 + 3) / 4)), dim3(256), 0, 0, A, n2, km);   // one column per wave, no loop
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const char *nm[4] = {"(i,j,k) i fastest, thread per column, k march", "(k,i,j) k innermost, wave per column, 2048 WGs grid-stride",
                         "(k,i,j) k innermost, wave per column, 8192 WGs grid-stride", "(k,i,j) k innermost, wave per column, one column per wave"};
    printf("%-62s %7.2f ms  %5.2f TB/s\n", nm[v], best, gb / best);
  }
  return 0;
}
