// mem_ceiling.hip -- what this MI355X sustains for the access pattern of the lane-per-filter step kernel.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mem_ceiling.hip -o /tmp/mem_ceiling ; run on the GPU box.
//   K0  out-of-place 16 B/lane copy (the guide's 6.29 TB/s reference)
//   K1  in-place read-modify-write stream, 16 B/lane, grid-stride
//   K2  per-wave contiguous 23 KB record tile -> LDS -> (lane-per-record regs) -> LDS -> same addresses, no math
//   K2c same with ~FMAS dependent-free fp64 FMAs per lane (stand-in for the filter algebra)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_copy(const double2* __restrict__ src, double2* __restrict__ dst, size_t nv) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nv; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

__global__ void k_rmw(double2* __restrict__ a, size_t nv) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nv; i += (size_t)gridDim.x * blockDim.x) {
    double2 v = a[i];
    v.x = v.x * 1.0000001 + 1e-9; v.y = v.y * 0.9999999 - 1e-9;
    a[i] = v;
  }
}

__device__ __forceinline__ void wsync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// EPF doubles per record, 64 records per wave-tile, one wave per block, in place
template <int EPF, int FMAS, int LANES>
__global__ __launch_bounds__(64) void k_tile(double* __restrict__ g, long ntiles) {
  __shared__ __attribute__((aligned(16))) double lds[LANES * EPF];
  const int lane = threadIdx.x;
  constexpr int NV = LANES * EPF / 2;
  constexpr int IT = (NV + 63) / 64;
  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    double2* g2 = reinterpret_cast<double2*>(g + tile * (long)(LANES * EPF));
    double2* l2 = reinterpret_cast<double2*>(lds);
    double2 v[IT];
#pragma unroll
    for (int i = 0; i < IT; i++) { int idx = lane + i * 64; if (idx < NV) v[i] = g2[idx]; }
#pragma unroll
    for (int i = 0; i < IT; i++) { int idx = lane + i * 64; if (idx < NV) l2[idx] = v[i]; }
    wsync();
    double r[EPF];
    const int ll = lane % LANES;
#pragma unroll
    for (int k = 0; k < EPF; k++) r[k] = lds[ll * EPF + k];
    if (FMAS > 0) {
#pragma unroll 1
      for (int rep = 0; rep < FMAS / EPF; rep++) {
#pragma unroll
        for (int k = 0; k < EPF; k++) r[k] = fma(r[k], 1.0000001, 1e-12);
      }
    }
    wsync();
    if (lane < LANES) {
#pragma unroll
      for (int k = 0; k < EPF; k++) lds[ll * EPF + k] = r[k];
    }
    wsync();
#pragma unroll
    for (int i = 0; i < IT; i++) { int idx = lane + i * 64; if (idx < NV) g2[idx] = l2[idx]; }
    wsync();
  }
}

template <class F>
double time_us(F launch, int reps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 20; i++) launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; i++) launch();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms * 1e3 / reps;
}

int main() {
  const long sizes[2] = {65536, 1048576};
  constexpr int EPF = 45;   // x(6) + P(36) + z(3) doubles per filter, moved in and out
  for (long n : sizes) {
    size_t bytes = (size_t)n * EPF * 8;
    double *a, *b;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes));
    size_t nv = bytes / 16;
    int reps = n > 100000 ? 50 : 400;
    printf("---- n = %ld filters, %.1f MB in + %.1f MB out per pass\n", n, bytes / 1e6, bytes / 1e6);
    for (int grid : {2048, 4096, 8192}) {
      double t0 = time_us([&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, (const double2*)a, (double2*)b, nv); }, reps);
      double t1 = time_us([&] { hipLaunchKernelGGL(k_rmw, dim3(grid), dim3(256), 0, 0, (double2*)a, nv); }, reps);
      printf("grid %5d x256: copy %8.2f us %6.2f TB/s | in-place rmw %8.2f us %6.2f TB/s\n", grid, t0, 2.0 * bytes / t0 / 1e6, t1, 2.0 * bytes / t1 / 1e6);
    }
    long tiles64 = n / 64, tiles32 = n / 32;
    for (int grid : {1024, 2048, 4096}) {
      long g64 = grid < tiles64 ? grid : tiles64;
      double t2 = time_us([&] { hipLaunchKernelGGL((k_tile<EPF, 0, 64>), dim3(g64), dim3(64), 0, 0, a, tiles64); }, reps);
      double t3 = time_us([&] { hipLaunchKernelGGL((k_tile<EPF, 450, 64>), dim3(g64), dim3(64), 0, 0, a, tiles64); }, reps);
      double t4 = time_us([&] { hipLaunchKernelGGL((k_tile<EPF, 900, 64>), dim3(g64), dim3(64), 0, 0, a, tiles64); }, reps);
      long g32 = 2 * grid < tiles32 ? 2 * grid : tiles32;
      double t5 = time_us([&] { hipLaunchKernelGGL((k_tile<EPF, 450, 32>), dim3(g32), dim3(64), 0, 0, a, tiles32); }, reps);
      printf("tile64 grid %5ld: nomath %8.2f us %5.2f TB/s | 450 fma %8.2f us %5.2f | 900 fma %8.2f us %5.2f || tile32 grid %5ld 450 fma %8.2f us %5.2f TB/s\n",
             g64, t2, 2.0 * bytes / t2 / 1e6, t3, 2.0 * bytes / t3 / 1e6, t4, 2.0 * bytes / t4 / 1e6, g32, t5, 2.0 * bytes / t5 / 1e6);
    }
    {
      // sensitivity to the record size (same total bytes is not kept: report TB/s)
      long t36 = (long)(bytes / 8 / (64 * 36)), t48 = (long)(bytes / 8 / (64 * 48)), t44 = (long)(bytes / 8 / (64 * 44));
      double a36 = time_us([&] { hipLaunchKernelGGL((k_tile<36, 0, 64>), dim3(t36 < 4096 ? t36 : 4096), dim3(64), 0, 0, a, t36); }, reps);
      double a44 = time_us([&] { hipLaunchKernelGGL((k_tile<44, 0, 64>), dim3(t44 < 4096 ? t44 : 4096), dim3(64), 0, 0, a, t44); }, reps);
      double a48 = time_us([&] { hipLaunchKernelGGL((k_tile<48, 0, 64>), dim3(t48 < 4096 ? t48 : 4096), dim3(64), 0, 0, a, t48); }, reps);
      printf("record size sweep (no math): EPF=36 %8.2f us %5.2f TB/s | EPF=44 %8.2f us %5.2f TB/s | EPF=48 %8.2f us %5.2f TB/s\n",
             a36, 2.0 * t36 * 64 * 36 * 8 / a36 / 1e6, a44, 2.0 * t44 * 64 * 44 * 8 / a44 / 1e6, a48, 2.0 * t48 * 64 * 48 * 8 / a48 / 1e6);
    }
    CK(hipFree(a)); CK(hipFree(b));
  }
  return 0;
}
