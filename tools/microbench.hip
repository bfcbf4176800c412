// microbench.hip -- device ceilings used by DESIGN.md / bench.py's roofline: fp64 stream copy,
// fp64 atomic-add stream, fp64 FMA rate.  Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_copy(double2 *__restrict__ d, const double2 *__restrict__ s, size_t n)
{
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    d[i] = s[i];
}
__global__ void k_atomic(double *__restrict__ d, const double *__restrict__ s, size_t n)
{
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    unsafeAtomicAdd(&d[i], s[i]);
}
__global__ void k_atomic_nosrc(double *__restrict__ d, size_t n)
{
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    unsafeAtomicAdd(&d[i], 1.0);
}
__global__ void k_rmw(double *__restrict__ d, const double *__restrict__ s, size_t n)
{
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    d[i] += s[i];
}
__global__ void k_fma(double *out, int iters)
{
  double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const double b = 1.0000001, c = 1e-9;
  for (int i = 0; i < iters; ++i)
    {
      a0 = fma(a0, b, c); a1 = fma(a1, b, c); a2 = fma(a2, b, c); a3 = fma(a3, b, c);
      a4 = fma(a4, b, c); a5 = fma(a5, b, c); a6 = fma(a6, b, c); a7 = fma(a7, b, c);
    }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

int main()
{
  const size_t n = (size_t)1 << 27; // 128 Mi doubles = 1 GiB
  double *a, *b;
  CK(hipMalloc(&a, n * 8));
  CK(hipMalloc(&b, n * 8));
  CK(hipMemset(a, 0, n * 8));
  CK(hipMemset(b, 0, n * 8));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms;
  auto time = [&](auto f, int reps) {
    f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
  };
  double t;
  t = time([&] { hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, 0, (double2 *)a, (const double2 *)b, n / 2); }, 10);
  printf("copy f64x2      : %.3f ms  %.2f TB/s (read+write)\n", t, 2.0 * n * 8 / t * 1e-9);
  t = time([&] { hipLaunchKernelGGL(k_rmw, dim3(2048), dim3(256), 0, 0, a, b, n); }, 10);
  printf("d += s (plain)  : %.3f ms  %.2f TB/s (3 streams)\n", t, 3.0 * n * 8 / t * 1e-9);
  t = time([&] { hipLaunchKernelGGL(k_atomic, dim3(2048), dim3(256), 0, 0, a, b, n); }, 5);
  printf("atomic f64 + src: %.3f ms  %.2f TB/s of added bytes\n", t, 1.0 * n * 8 / t * 1e-9);
  t = time([&] { hipLaunchKernelGGL(k_atomic_nosrc, dim3(2048), dim3(256), 0, 0, a, n); }, 5);
  printf("atomic f64 only : %.3f ms  %.2f TB/s of added bytes\n", t, 1.0 * n * 8 / t * 1e-9);
  t = time([&] { hipMemsetAsync(a, 0, n * 8, 0); }, 10);
  printf("memset          : %.3f ms  %.2f TB/s\n", t, 1.0 * n * 8 / t * 1e-9);
  const int iters = 4096;
  t = time([&] { hipLaunchKernelGGL(k_fma, dim3(256 * 8), dim3(256), 0, 0, a, iters); }, 5);
  printf("fma f64         : %.3f ms  %.2f TFLOP/s\n", t, 2.0 * 8 * iters * 256.0 * 8 * 256 / t * 1e-9);
  return 0;
}
