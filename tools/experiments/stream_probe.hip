// stream_probe.hip -- round 4: what the memory system sustains for the access shapes a brick kernel can choose
// between (per-lane width, loads in flight, waves per CU, persistent brick-shaped streams with LDS staging).
// Build: hipcc --offload-arch=gfx950 -O3 -o stream_probe.bin stream_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x)                                                         \
  do                                                                  \
    {                                                                 \
      hipError_t err_ = (x);                                             \
      if (err_ != hipSuccess)                                            \
        {                                                             \
          printf("%s: %s\n", #x, hipGetErrorString(err_));               \
          exit(1);                                                    \
        }                                                             \
    }                                                                 \
  while (0)

template <int W>
struct Vec;
template <>
struct Vec<8>
{
  using type = double;
};
template <>
struct Vec<16>
{
  typedef double type __attribute__((ext_vector_type(2)));
};

// grid-stride copy, W bytes per lane, U loads in flight per lane, PAD bytes of LDS to cap the residency
template <int W, int U, int PAD, bool NT>
__global__ void __launch_bounds__(256) k_copy(typename Vec<W>::type *__restrict__ d, const typename Vec<W>::type *__restrict__ s,
                                              size_t n)
{
  using V = typename Vec<W>::type;
  __shared__ char pad[PAD > 0 ? PAD : 1];
  if (PAD > 0 && n == 0)
    pad[threadIdx.x] = 1;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t       i      = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride)
    {
      V v[U];
#pragma unroll
      for (int u = 0; u < U; ++u)
        v[u] = NT ? __builtin_nontemporal_load(&s[i + u * stride]) : s[i + u * stride];
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (NT)
          __builtin_nontemporal_store(v[u], &d[i + u * stride]);
        else
          d[i + u * stride] = v[u];
    }
  for (; i < n; i += stride)
    d[i] = s[i];
}

template <int W, int U>
__global__ void __launch_bounds__(256) k_read(double *__restrict__ out, const typename Vec<W>::type *__restrict__ s, size_t n)
{
  using V = typename Vec<W>::type;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t       i      = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  double       acc    = 0;
  for (; i + (U - 1) * stride < n; i += U * stride)
    {
      V v[U];
#pragma unroll
      for (int u = 0; u < U; ++u)
        v[u] = s[i + u * stride];
#pragma unroll
      for (int u = 0; u < U; ++u)
        acc += *reinterpret_cast<const double *>(&v[u]);
    }
  if (acc == 1.2345e-300)
    out[0] = acc;
}

template <int W>
__global__ void __launch_bounds__(256) k_write(typename Vec<W>::type *__restrict__ d, size_t n)
{
  using V = typename Vec<W>::type;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  V            v;
  *reinterpret_cast<double *>(&v) = 1.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    d[i] = v;
}

// persistent brick-shaped stream: per brick NIN doubles in (contiguous from brick * NOUT: the 17^3 / 16^3 halo overlap),
// staged through LDS, optional pseudo-sweeps, NOUT doubles out.  Software pipeline of the macro-element kernel: the loads
// of brick n + 1 are issued before the stores of brick n.  NS streams read per brick (the Chebyshev forms read x, b,
// x_old: NS = 3), LDSB bytes of LDS per workgroup (residency).
template <int W, int NS, int SWEEP, int LDSB, int NT_>
__global__ void __launch_bounds__(NT_) k_brick(double *__restrict__ d, const double *__restrict__ s0, const double *__restrict__ s1,
                                               const double *__restrict__ s2, uint32_t n_bricks)
{
  constexpr int NIN = 4913, NOUT = 4096, NT = NT_, E = W / 8;
  constexpr int JI = (NIN / E + NT - 1) / NT, JO = (NOUT / E + NT - 1) / NT;
  extern __shared__ double lds[];
  double                  *U = lds, *Wd = LDSB >= 2 * 4913 * 8 ? lds + 4913 : lds; // (small LDSB: aliased, values are garbage anyway)
  using V                    = typename Vec<W>::type;
  const int tid              = threadIdx.x;
  V         g[JI];
  auto      issue = [&](uint32_t b) {
    const V *p = reinterpret_cast<const V *>(s0 + (size_t)b * NOUT);
#pragma unroll
    for (int j = 0; j < JI; ++j)
      {
        const int i = tid + j * NT;
        if (i < NIN / E)
          g[j] = p[i];
      }
  };
  auto land = [&]() {
#pragma unroll
    for (int j = 0; j < JI; ++j)
      {
        const int i = tid + j * NT;
        if (i < NIN / E)
          reinterpret_cast<V *>(U)[i] = g[j];
      }
  };
  uint32_t b = blockIdx.x;
  if (b >= n_bricks)
    return;
  issue(b);
  land();
  __syncthreads();
  for (;;)
    {
      const uint32_t bn = b + gridDim.x;
      const bool     hn = bn < n_bricks;
      // pseudo-sweeps: SWEEP passes, each reads a 17-line of U and W, ~100 FMAs, writes back (three barriers)
#pragma unroll 1
      for (int sw = 0; sw < SWEEP; ++sw)
        {
          double a[17], acc[17];
#pragma unroll
          for (int j = 0; j < 17; ++j)
            a[j] = U[(tid * 17 + j) % NIN];
#pragma unroll
          for (int j = 0; j < 17; ++j)
            {
              acc[j] = a[j];
#pragma unroll
              for (int k = 0; k < 5; ++k)
                acc[j] = fma(a[(j + k) % 17], 1.0000001, acc[j]);
            }
#pragma unroll
          for (int j = 0; j < 17; ++j)
            {
              Wd[(tid * 17 + j) % NIN] = acc[j];
              U[(tid * 17 + j) % NIN]  = acc[(j + 1) % 17];
            }
          __syncthreads();
        }
      if (SWEEP == 0)
        {
#pragma unroll
          for (int j = 0; j < JI; ++j)
            {
              const int i = tid + j * NT;
              if (i < NIN / E)
                reinterpret_cast<V *>(Wd)[i] = reinterpret_cast<V *>(U)[i];
            }
          __syncthreads();
        }
      if (hn)
        issue(bn);
      // write-out: NS - 1 further operand streams read at the output points, one stream written
      V *o = reinterpret_cast<V *>(d + (size_t)b * NOUT);
      const V *p1 = reinterpret_cast<const V *>(s1 + (size_t)b * NOUT), *p2 = reinterpret_cast<const V *>(s2 + (size_t)b * NOUT);
      constexpr int CH = 4;
#pragma unroll
      for (int j0 = 0; j0 < JO; j0 += CH)
        {
          V x1[CH], x2[CH];
#pragma unroll
          for (int c = 0; c < CH; ++c)
            {
              const int i = tid + (j0 + c) * NT;
              if (j0 + c < JO && i < NOUT / E)
                {
                  if (NS >= 2)
                    x1[c] = p1[i];
                  if (NS >= 3)
                    x2[c] = p2[i];
                }
            }
#pragma unroll
          for (int c = 0; c < CH; ++c)
            {
              const int i = tid + (j0 + c) * NT;
              if (j0 + c < JO && i < NOUT / E)
                {
                  V v = reinterpret_cast<V *>(Wd)[i];
                  if (NS >= 2)
                    *reinterpret_cast<double *>(&v) += *reinterpret_cast<double *>(&x1[c]);
                  if (NS >= 3)
                    *reinterpret_cast<double *>(&v) += *reinterpret_cast<double *>(&x2[c]);
                  o[i] = v;
                }
            }
        }
      if (!hn)
        break;
      __syncthreads();
      land();
      __syncthreads();
      b = bn;
    }
}


// The same stream with DEEP prefetch: all global loads of brick n + 1 (source and, NS = 3, the two operand streams of
// the write-out) are issued BEFORE the pseudo-sweeps of brick n and land in registers after them; the write-out of
// brick n has every operand in registers and waits for nothing.  WPS = waves per SIMD the register allocation is
// asked to allow (1: one workgroup per CU with up to 512 registers; 2: two workgroups per CU, 256 registers).
template <int NS, int SWEEP, int WPS>
__global__ void __launch_bounds__(256, WPS) k_brick_deep(double *__restrict__ d, const double *__restrict__ s0,
                                                         const double *__restrict__ s1, const double *__restrict__ s2,
                                                         uint32_t n_bricks)
{
  constexpr int NIN = 4913, NOUT = 4096, NT = 256;
  constexpr int JI = (NIN + NT - 1) / NT, JO = NOUT / NT;
  extern __shared__ double lds[];
  double                  *U = lds, *Wd = lds + 4913;
  const int tid              = threadIdx.x;
  double    g[JI], x1[NS >= 2 ? JO : 1], x2[NS >= 3 ? JO : 1];
  auto      issue = [&](uint32_t b) {
    const double *p = s0 + (size_t)b * NOUT, *p1 = s1 + (size_t)b * NOUT, *p2 = s2 + (size_t)b * NOUT;
#pragma unroll
    for (int j = 0; j < JI; ++j)
      {
        const int i = tid + j * NT;
        g[j]        = i < NIN ? p[i] : 0.;
      }
#pragma unroll
    for (int j = 0; j < JO; ++j)
      {
        if (NS >= 2)
          x1[j] = p1[tid + j * NT];
        if (NS >= 3)
          x2[NS >= 3 ? j : 0] = p2[tid + j * NT];
      }
  };
  uint32_t b = blockIdx.x;
  if (b >= n_bricks)
    return;
  issue(b);
  // (prologue: the first brick's operands are consumed right away)
  double o1[NS >= 2 ? JO : 1], o2[NS >= 3 ? JO : 1];
  for (;;)
    {
#pragma unroll
      for (int j = 0; j < JI; ++j)
        if (tid + j * NT < NIN)
          U[tid + j * NT] = g[j];
#pragma unroll
      for (int j = 0; j < JO; ++j)
        {
          if (NS >= 2)
            o1[j] = x1[j];
          if (NS >= 3)
            o2[NS >= 3 ? j : 0] = x2[NS >= 3 ? j : 0];
        }
      __syncthreads();
      const uint32_t bn = b + gridDim.x;
      const bool     hn = bn < n_bricks;
      if (hn)
        issue(bn); // in flight during the sweeps
#pragma unroll 1
      for (int sw = 0; sw < SWEEP; ++sw)
        {
          double a[17], acc[17];
#pragma unroll
          for (int j = 0; j < 17; ++j)
            a[j] = U[(tid * 17 + j) % NIN];
#pragma unroll
          for (int j = 0; j < 17; ++j)
            {
              acc[j] = a[j];
#pragma unroll
              for (int k = 0; k < 5; ++k)
                acc[j] = fma(a[(j + k) % 17], 1.0000001, acc[j]);
            }
#pragma unroll
          for (int j = 0; j < 17; ++j)
            {
              Wd[(tid * 17 + j) % NIN] = acc[j];
              U[(tid * 17 + j) % NIN]  = acc[(j + 1) % 17];
            }
          __syncthreads();
        }
      double *o = d + (size_t)b * NOUT;
#pragma unroll
      for (int j = 0; j < JO; ++j)
        {
          double v = Wd[tid + j * NT];
          if (NS >= 2)
            v += o1[j];
          if (NS >= 3)
            v += o2[NS >= 3 ? j : 0];
          o[tid + j * NT] = v;
        }
      if (!hn)
        break;
      __syncthreads();
      b = bn;
    }
}

int main()
{
  const size_t n = (size_t)1 << 27; // 128 Mi doubles = 1 GiB per array
  double      *a, *b, *c, *e;
  CK(hipMalloc(&a, n * 8 + 65536));
  CK(hipMalloc(&b, n * 8 + 65536));
  CK(hipMalloc(&c, n * 8 + 65536));
  CK(hipMalloc(&e, n * 8 + 65536));
  CK(hipMemset(a, 0, n * 8));
  CK(hipMemset(b, 0, n * 8));
  CK(hipMemset(c, 0, n * 8));
  CK(hipMemset(e, 0, n * 8));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto time = [&](auto f, int reps) {
    float ms;
    f();
    f();
    CK(hipDeviceSynchronize());
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r)
      f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    CK(hipGetLastError());
    hipEventElapsedTime(&ms, e0, e1);
    return (double)ms / reps;
  };
  double t;
#define COPY(W, U, PAD, NT, BLOCKS)                                                                                           \
  t = time([&] { hipLaunchKernelGGL((k_copy<W, U, PAD, NT>), dim3(BLOCKS), dim3(256), 0, 0, (Vec<W>::type *)a,                 \
                                    (const Vec<W>::type *)b, n * 8 / W); },                                                    \
           10);                                                                                                               \
  printf("copy  W=%2d U=%d ldspad=%6d nt=%d blocks=%5d : %.3f ms  %.2f TB/s\n", W, U, PAD, NT, BLOCKS, t, 2.0 * n * 8 / t * 1e-9);
  COPY(8, 1, 0, false, 2048)
  COPY(16, 1, 0, false, 2048)
  COPY(8, 4, 0, false, 2048)
  COPY(16, 4, 0, false, 2048)
  COPY(8, 8, 0, false, 2048)
  COPY(16, 8, 0, false, 2048)
  COPY(16, 4, 0, true, 2048)
  COPY(16, 8, 0, true, 2048)
  COPY(8, 8, 0, true, 2048)
  COPY(16, 4, 0, false, 8192)
  COPY(16, 4, 0, false, 512)
  COPY(16, 8, 0, false, 512)
  COPY(8, 8, 0, false, 512)
  COPY(8, 16, 0, false, 512)
  COPY(16, 16, 0, false, 512)
  // residency capped by LDS: 2 workgroups (8 waves) per CU, as the macro-element kernel
  COPY(8, 8, 65000, false, 512)
  COPY(8, 16, 65000, false, 512)
  COPY(16, 8, 65000, false, 512)
  COPY(16, 16, 65000, false, 512)
  COPY(16, 8, 65000, true, 512)
  COPY(8, 16, 65000, true, 512)
  // 4 workgroups per CU
  COPY(8, 8, 40000, false, 1024)
  COPY(16, 8, 40000, false, 1024)
#define READ(W, U, BLOCKS)                                                                                                    \
  t = time([&] { hipLaunchKernelGGL((k_read<W, U>), dim3(BLOCKS), dim3(256), 0, 0, a, (const Vec<W>::type *)b, n * 8 / W); }, 10); \
  printf("read  W=%2d U=%d blocks=%5d : %.3f ms  %.2f TB/s\n", W, U, BLOCKS, t, 1.0 * n * 8 / t * 1e-9);
  READ(8, 8, 2048)
  READ(16, 8, 2048)
  READ(8, 16, 512)
  READ(16, 16, 512)
#define WRITE(W, BLOCKS)                                                                                                      \
  t = time([&] { hipLaunchKernelGGL((k_write<W>), dim3(BLOCKS), dim3(256), 0, 0, (Vec<W>::type *)a, n * 8 / W); }, 10);         \
  printf("write W=%2d blocks=%5d : %.3f ms  %.2f TB/s\n", W, BLOCKS, t, 1.0 * n * 8 / t * 1e-9);
  WRITE(8, 2048)
  WRITE(16, 2048)

  const uint32_t nbr = 32768 - 1; // bricks of 4096 outputs (the last one would read beyond the array)
#define BRICK(W, NS, SWEEP, LDSB, NT, WGS)                                                                                    \
  {                                                                                                                           \
    CK(hipFuncSetAttribute((const void *)k_brick<W, NS, SWEEP, LDSB, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));  \
    t = time([&] { hipLaunchKernelGGL((k_brick<W, NS, SWEEP, LDSB, NT>), dim3(256 * WGS), dim3(NT), LDSB, 0, a, b, c, e, nbr); }, 5); \
    const double bytes = (double)nbr * (4913 + (NS - 1) * 4096 + 4096) * 8;                                                    \
    printf("brick W=%2d streams_in=%d sweeps=%d lds=%6d threads=%d wg/cu=%d : %.3f ms  %.2f TB/s actual (%.2f TB/s of %d B/DoF)\n", \
           W, NS, SWEEP, LDSB, NT, WGS, t, bytes / t * 1e-9, (double)nbr * 4096 * (NS + 1) * 8 / t * 1e-9, 8 * (NS + 1));       \
  }
  BRICK(8, 1, 0, 78608, 256, 2)
  BRICK(16, 1, 0, 78608, 256, 2)
  BRICK(8, 1, 3, 78608, 256, 2)
  BRICK(16, 1, 3, 78608, 256, 2)
  BRICK(8, 1, 6, 78608, 256, 2)
  BRICK(8, 3, 0, 78608, 256, 2)
  BRICK(16, 3, 0, 78608, 256, 2)
  BRICK(8, 3, 3, 78608, 256, 2)
  BRICK(16, 3, 3, 78608, 256, 2)
  // one workgroup of 512 threads per CU with twice the LDS / two per CU
  BRICK(8, 1, 3, 78608, 512, 2)
  BRICK(16, 1, 3, 78608, 512, 2)
  BRICK(8, 3, 3, 78608, 512, 2)
  // more workgroups per CU (what a smaller LDS footprint would buy): 3 and 4
  BRICK(8, 1, 3, 78608 / 2 + 8, 256, 4)
  BRICK(16, 1, 3, 78608 / 2 + 8, 256, 4)
  BRICK(8, 3, 3, 78608 / 2 + 8, 256, 4)
  BRICK(16, 3, 3, 78608 / 2 + 8, 256, 4)
#define DEEP(NS, SWEEP, WPS)                                                                                                  \
  {                                                                                                                           \
    constexpr int LDSB = WPS == 1 ? 150000 : 78608;                                                                           \
    CK(hipFuncSetAttribute((const void *)k_brick_deep<NS, SWEEP, WPS>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));     \
    t = time([&] { hipLaunchKernelGGL((k_brick_deep<NS, SWEEP, WPS>), dim3(256 * WPS), dim3(256), LDSB, 0, a, b, c, e, nbr); }, 5); \
    const double bytes = (double)nbr * (4913 + (NS - 1) * 4096 + 4096) * 8;                                                    \
    printf("deep  streams_in=%d sweeps=%d wg/cu=%d : %.3f ms  %.2f TB/s actual (%.2f TB/s of %d B/DoF)\n", NS, SWEEP, WPS, t, \
           bytes / t * 1e-9, (double)nbr * 4096 * (NS + 1) * 8 / t * 1e-9, 8 * (NS + 1));                                      \
  }
  DEEP(1, 3, 2)
  DEEP(1, 6, 2)
  DEEP(1, 9, 2)
  DEEP(3, 3, 2)
  DEEP(3, 6, 2)
  DEEP(3, 9, 2)
  DEEP(1, 3, 1)
  DEEP(1, 6, 1)
  DEEP(1, 9, 1)
  DEEP(3, 3, 1)
  DEEP(3, 6, 1)
  DEEP(3, 9, 1)
  BRICK(8, 1, 9, 78608, 256, 2)
  BRICK(8, 3, 6, 78608, 256, 2)
  BRICK(8, 3, 9, 78608, 256, 2)
  return 0;
}
