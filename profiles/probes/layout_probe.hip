// layout_probe.hip -- does the level stride of the (i,j,k) layout limit the k-marching stencil kernels?
// A workgroup owns a 64x8 tile of columns and marches 62 levels, reading NA arrays and writing 2 per level, exactly
// the access shape of k_momentum_rhs_lds (no LDS, no arithmetic).  Layout 0: (i,j,k) -- level stride = nx*ny
// (69 MB at tx0.1v3); layout 1: (i,k,j) -- level stride = nx (29 KB).  Same bytes, same coalescing.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
#define NA 10
struct Arrs { const double *a[NA]; double *o[2]; };
template <int LAYOUT>
__global__ void __launch_bounds__(512) probe(Arrs A, int nx, int ny, int km, int tiles_i) {
  const int ti = blockIdx.x % tiles_i, tj = blockIdx.x / tiles_i;
  const int i = ti * 64 + threadIdx.x, j = tj * 8 + threadIdx.y;
  if (i >= nx || j >= ny) return;
  const size_t n2 = (size_t)nx * ny;
  double nxt[NA];
  auto idx = [&](int k) { return LAYOUT == 0 ? (size_t)k * n2 + (size_t)j * nx + i : ((size_t)j * km + k) * nx + i; };
#pragma unroll
  for (int a = 0; a < NA; ++a) nxt[a] = A.a[a][idx(0)];
  for (int k = 0; k < km; ++k) {
    double cur[NA];
#pragma unroll
    for (int a = 0; a < NA; ++a) cur[a] = nxt[a];
    const int kn = k + 1 < km ? k + 1 : k;
#pragma unroll
    for (int a = 0; a < NA; ++a) nxt[a] = A.a[a][idx(kn)];
    double s0 = 0, s1 = 0;
#pragma unroll
    for (int a = 0; a < NA; a += 2) { s0 += cur[a]; s1 += cur[a + 1]; }
    A.o[0][idx(k)] = s0; A.o[1][idx(k)] = s1;
  }
}
// plain linear pass over the same 10 + 2 arrays: every thread one cell (the ceiling for this stream mix)
__global__ void __launch_bounds__(256) linear(Arrs A, size_t n) {
  const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= n) return;
  double s0 = 0, s1 = 0;
#pragma unroll
  for (int a = 0; a < NA; a += 2) { s0 += A.a[a][q]; s1 += A.a[a + 1][q]; }
  A.o[0][q] = s0; A.o[1][q] = s1;
}
// level-chunked 3-D parallel form: thread = one column, KC consecutive levels; grid (tiles, km/KC)
template <int KC>
__global__ void __launch_bounds__(256) chunked(Arrs A, int nx, int ny, int km) {
  const size_t n2 = (size_t)nx * ny;
  const size_t p2 = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (p2 >= n2) return;
  const int k0 = blockIdx.y * KC;
  for (int k = k0; k < k0 + KC && k < km; ++k) {
    const size_t q = (size_t)k * n2 + p2;
    double s0 = 0, s1 = 0;
#pragma unroll
    for (int a = 0; a < NA; a += 2) { s0 += A.a[a][q]; s1 += A.a[a + 1][q]; }
    A.o[0][q] = s0; A.o[1][q] = s1;
  }
}
// column march with 64-thread workgroups (one wave per WG, natural order): the shape of the register Thomas kernels
__global__ void __launch_bounds__(64) colmarch(Arrs A, int nx, int ny, int km) {
  const size_t n2 = (size_t)nx * ny;
  const size_t p2 = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (p2 >= n2) return;
  double nxt[NA];
#pragma unroll
  for (int a = 0; a < NA; ++a) nxt[a] = A.a[a][p2];
  for (int k = 0; k < km; ++k) {
    double cur[NA];
#pragma unroll
    for (int a = 0; a < NA; ++a) cur[a] = nxt[a];
    const int kn = k + 1 < km ? k + 1 : k;
#pragma unroll
    for (int a = 0; a < NA; ++a) nxt[a] = A.a[a][(size_t)kn * n2 + p2];
    double s0 = 0, s1 = 0;
#pragma unroll
    for (int a = 0; a < NA; a += 2) { s0 += cur[a]; s1 += cur[a + 1]; }
    A.o[0][(size_t)k * n2 + p2] = s0; A.o[1][(size_t)k * n2 + p2] = s1;
  }
}
// tile march, TW x TH threads per workgroup, natural or XCD patch order
template <int TW, int TH>
__global__ void __launch_bounds__(TW * TH) tilemarch(Arrs A, int nx, int ny, int km, int tiles_i) {
  const int ti = blockIdx.x % tiles_i, tj = blockIdx.x / tiles_i;
  const int i = ti * TW + threadIdx.x, j = tj * TH + threadIdx.y;
  if (i >= nx || j >= ny) return;
  const size_t n2 = (size_t)nx * ny, p2 = (size_t)j * nx + i;
  double nxt[NA];
#pragma unroll
  for (int a = 0; a < NA; ++a) nxt[a] = A.a[a][p2];
  for (int k = 0; k < km; ++k) {
    double cur[NA];
#pragma unroll
    for (int a = 0; a < NA; ++a) cur[a] = nxt[a];
    const int kn = k + 1 < km ? k + 1 : k;
#pragma unroll
    for (int a = 0; a < NA; ++a) nxt[a] = A.a[a][(size_t)kn * n2 + p2];
    double s0 = 0, s1 = 0;
#pragma unroll
    for (int a = 0; a < NA; a += 2) { s0 += cur[a]; s1 += cur[a + 1]; }
    A.o[0][(size_t)k * n2 + p2] = s0; A.o[1][(size_t)k * n2 + p2] = s1;
  }
}
// tile march with two adjacent columns per thread (16-byte loads): TW x TH threads cover 2*TW x TH columns
template <int TW, int TH>
__global__ void __launch_bounds__(TW * TH) tilemarch2(Arrs A, int nx, int ny, int km, int tiles_i) {
  const int ti = blockIdx.x % tiles_i, tj = blockIdx.x / tiles_i;
  const int i = ti * 2 * TW + 2 * threadIdx.x, j = tj * TH + threadIdx.y;
  if (i + 1 >= nx || j >= ny) return;
  const size_t n2 = (size_t)nx * ny, p2 = (size_t)j * nx + i;
  double2 nxt[NA];
#pragma unroll
  for (int a = 0; a < NA; ++a) nxt[a] = *reinterpret_cast<const double2 *>(A.a[a] + p2);
  for (int k = 0; k < km; ++k) {
    double2 cur[NA];
#pragma unroll
    for (int a = 0; a < NA; ++a) cur[a] = nxt[a];
    const int kn = k + 1 < km ? k + 1 : k;
#pragma unroll
    for (int a = 0; a < NA; ++a) nxt[a] = *reinterpret_cast<const double2 *>(A.a[a] + (size_t)kn * n2 + p2);
    double2 s0 = make_double2(0, 0), s1 = make_double2(0, 0);
#pragma unroll
    for (int a = 0; a < NA; a += 2) { s0.x += cur[a].x; s0.y += cur[a].y; s1.x += cur[a + 1].x; s1.y += cur[a + 1].y; }
    *reinterpret_cast<double2 *>(A.o[0] + (size_t)k * n2 + p2) = s0;
    *reinterpret_cast<double2 *>(A.o[1] + (size_t)k * n2 + p2) = s1;
  }
}
int main() {
  const int nx = (getenv("NX") ? atoi(getenv("NX")) : 3604), ny = 2404, km = 62;
  const size_t n = (size_t)nx * ny * km;
  Arrs A;
  // STAGGER=<bytes>: array a starts a*STAGGER bytes into its allocation (does the relative placement of the streams matter?)
  const size_t stag = getenv("STAGGER") ? (size_t)atol(getenv("STAGGER")) / 8 : 0;
  for (int a = 0; a < NA; ++a) { double *p; if (hipMalloc(&p, (n + 16 * stag) * 8) != hipSuccess) { printf("alloc failed\n"); return 1; } hipMemset(p, 0, (n + 16 * stag) * 8); A.a[a] = p + a * stag; }
  for (int a = 0; a < 2; ++a) { double *p; hipMalloc(&p, (n + 16 * stag) * 8); A.o[a] = p + (NA + a) * stag; }
  const int tiles_i = (nx + 63) / 64, tiles_j = (ny + 7) / 8;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int layout = 0; layout < 2; ++layout) {
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
      hipEventRecord(e0);
      if (layout == 0) hipLaunchKernelGGL(probe<0>, dim3(tiles_i * tiles_j), dim3(64, 8), 0, 0, A, nx, ny, km, tiles_i);
      else hipLaunchKernelGGL(probe<1>, dim3(tiles_i * tiles_j), dim3(64, 8), 0, 0, A, nx, ny, km, tiles_i);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("layout %d (%s): %.3f ms, %.2f TB/s\n", layout, layout ? "i,k,j" : "i,j,k", best, (NA + 2) * n * 8 / (best * 1e-3) / 1e12);
  }
  auto timeit = [&](const char *name, auto launch) {
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
      hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("%-28s %.3f ms, %.2f TB/s\n", name, best, (NA + 2) * n * 8 / (best * 1e-3) / 1e12);
  };
  const size_t n2 = (size_t)nx * ny;
  timeit("linear 1 cell/thread", [&] { hipLaunchKernelGGL(linear, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, A, n); });
  timeit("chunked 8 levels/thread", [&] { hipLaunchKernelGGL(chunked<8>, dim3((unsigned)((n2 + 255) / 256), (km + 7) / 8), dim3(256), 0, 0, A, nx, ny, km); });
  timeit("chunked 16 levels/thread", [&] { hipLaunchKernelGGL(chunked<16>, dim3((unsigned)((n2 + 255) / 256), (km + 15) / 16), dim3(256), 0, 0, A, nx, ny, km); });
  timeit("column march, 64-thread WGs", [&] { hipLaunchKernelGGL(colmarch, dim3((unsigned)((n2 + 63) / 64)), dim3(64), 0, 0, A, nx, ny, km); });
#define TM(TW, TH) timeit("tile march " #TW "x" #TH, [&] { const int tI = (nx + TW - 1) / TW, tJ = (ny + TH - 1) / TH; \
    hipLaunchKernelGGL((tilemarch<TW, TH>), dim3(tI * tJ), dim3(TW, TH), 0, 0, A, nx, ny, km, tI); });
  TM(64, 1) TM(64, 2) TM(64, 4) TM(64, 8) TM(64, 16) TM(128, 1) TM(256, 1) TM(512, 1) TM(128, 4) TM(256, 2)
#define TM2(TW, TH) timeit("tile march2 (16B) " #TW "x" #TH, [&] { const int tI = (nx + 2 * TW - 1) / (2 * TW), tJ = (ny + TH - 1) / TH; \
    hipLaunchKernelGGL((tilemarch2<TW, TH>), dim3(tI * tJ), dim3(TW, TH), 0, 0, A, nx, ny, km, tI); });
  TM2(64, 1) TM2(64, 4) TM2(64, 8) TM2(32, 8) TM2(32, 16)
  return 0;
}
