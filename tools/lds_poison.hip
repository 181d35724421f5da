// Fills the LDS of every CU with NaN bit patterns (debugging aid: makes reads of LDS a kernel never wrote show up as NaN
// instead of whatever the previous kernel left there).  extern "C" void lds_poison(void* stream)
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(256) void k_poison(double* sink) {
  __shared__ double s[8192];                 // 64 KB
  for (int i = threadIdx.x; i < 8192; i += 256) s[i] = __longlong_as_double(0x7ff8dead0000beefLL);
  __syncthreads();
  if (sink != nullptr && s[threadIdx.x] == 0.0) sink[0] = 1.0;      // keep the stores alive
}

extern "C" void lds_poison(void* stream) {
  for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(k_poison, dim3(2048), dim3(256), 0, (hipStream_t)stream, (double*)nullptr);
}
