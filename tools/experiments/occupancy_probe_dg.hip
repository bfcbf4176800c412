// How many workgroups of a given LDS footprint / thread count / register budget does one CU of the
// MI355X take at once?  Every workgroup spins for a fixed number of cycles; 2 x CUs workgroups are
// launched: the launch lasts one spin if two are resident per CU, two spins otherwise.
// build: hipcc --offload-arch=gfx950 -O3 tools/experiments/occupancy_probe.hip -o gpurun_out/occupancy_probe
#include <hip/hip_runtime.h>
#include <cstdio>

template <int LDS_BYTES, int THREADS, int MINW>
__global__ void __launch_bounds__(THREADS, MINW) spin(unsigned long long cycles, double *out)
{
  __shared__ double buf[LDS_BYTES / 8];
  buf[threadIdx.x] = threadIdx.x;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < cycles)
    __builtin_amdgcn_s_sleep(8);
  if (buf[(threadIdx.x + 1) % THREADS] < 0)
    out[0] = 1;
}

template <int LDS_BYTES, int THREADS, int MINW>
void probe(int wgs)
{
  double *out;
  hipMalloc(&out, 8);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  int nb = 0;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, spin<LDS_BYTES, THREADS, MINW>, THREADS, 0);
  spin<LDS_BYTES, THREADS, MINW><<<wgs, THREADS>>>(1000, out);
  hipDeviceSynchronize();
  hipEventRecord(a);
  spin<LDS_BYTES, THREADS, MINW><<<wgs, THREADS>>>(240000, out); // ~100 us at 2.4 GHz
  hipEventRecord(b);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  printf("LDS %6d B threads %4d minw %d wgs %4d: API blocks/CU %d, launch %.1f us (one spin = ~100 us)\n", LDS_BYTES, THREADS, MINW,
         wgs, nb, ms * 1e3);
  hipFree(out);
}

int main()
{
  // how many 256-thread workgroups of the DG kernel's LDS footprint (39.4 kB) are resident per CU?
  probe<39936, 256, 1>(512);
  probe<39936, 256, 1>(768);
  probe<39936, 256, 1>(1024);
  probe<39936, 256, 1>(1280);
  probe<24576, 256, 1>(1024);
  probe<24576, 256, 1>(1536);
  probe<24576, 256, 1>(2048);
  probe<19968, 128, 1>(2048);
  probe<19968, 128, 1>(4096);
  return 0;
}
