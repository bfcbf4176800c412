// mgx_vector.hip -- vector kernels of the multigrid driver (SURVEY.md 8a row U and the
// PreconditionChebyshev / SolverCG vector updates).  All are streaming, HBM-bound kernels:
// grid-stride loops over a grid capped at 256 CUs x 8 workgroups, one element per lane and
// iteration (fp64: 512 B per wave instruction).
#include "mgx_internal.hpp"

#include <hip/hip_runtime.h>

namespace mgx
{
  static inline dim3 stream_grid(size_t n)
  {
    size_t nb = (n + 255) / 256;
    if (nb > 2048)
      nb = 2048;
    if (nb < 1)
      nb = 1;
    return dim3((unsigned)nb);
  }

#define GRID_STRIDE(i, n) \
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (size_t)gridDim.x * blockDim.x)

  template <typename TD, typename TS>
  __global__ void __launch_bounds__(256) k_copy_cast(TD *__restrict__ dst, const TS *__restrict__ src, size_t n)
  {
    GRID_STRIDE(i, n) dst[i] = (TD)src[i];
  }

  template <typename TD, typename TS>
  __global__ void __launch_bounds__(256) k_add_cast(TD *__restrict__ dst, const TS *__restrict__ src, size_t n)
  {
    GRID_STRIDE(i, n) dst[i] += (TD)src[i];
  }

  template <typename T>
  __global__ void __launch_bounds__(256) k_sadd(T *__restrict__ x, T sx, T a, const T *__restrict__ v, size_t n)
  {
    GRID_STRIDE(i, n) x[i] = sx * x[i] + a * v[i];
  }

  template <typename T>
  __global__ void __launch_bounds__(256) k_rhs_minus(T *__restrict__ res, const T *__restrict__ rhs, size_t n)
  {
    GRID_STRIDE(i, n) res[i] = rhs[i] - res[i];
  }

  template <typename T>
  __global__ void __launch_bounds__(256)
    k_constrained_copy(T *__restrict__ dst, const T *__restrict__ src, const uint32_t *__restrict__ list,
                       uint32_t count)
  {
    GRID_STRIDE(i, count)
    {
      const uint32_t c = list[i];
      dst[c]           = src[c];
    }
  }

  template <typename T>
  __global__ void __launch_bounds__(256)
    k_constrained_residual(T *__restrict__ res, const T *__restrict__ rhs, const T *__restrict__ lhs,
                           const uint32_t *__restrict__ list, uint32_t count)
  {
    GRID_STRIDE(i, count)
    {
      const uint32_t c = list[i];
      res[c]           = rhs[c] - lhs[c];
    }
  }

  template <typename T>
  __global__ void __launch_bounds__(256)
    k_constrained_set(T *__restrict__ v, T value, const uint32_t *__restrict__ list, uint32_t count)
  {
    GRID_STRIDE(i, count) v[list[i]] = value;
  }

  template <typename T>
  __global__ void __launch_bounds__(256) k_invert(T *__restrict__ v, size_t n)
  {
    GRID_STRIDE(i, n) v[i] = T(1) / v[i];
  }

  template <typename T>
  __global__ void __launch_bounds__(256)
    k_scatter_values(T *__restrict__ v, const uint32_t *__restrict__ idx, const double *__restrict__ val,
                     uint32_t count)
  {
    GRID_STRIDE(i, count) v[idx[i]] = (T)val[i];
  }

  template <typename T, int MODE>
  __global__ void __launch_bounds__(256)
    k_cheb_update(T *__restrict__ x, T *__restrict__ x_old, const T *__restrict__ b, const T *__restrict__ t,
                  const T *__restrict__ dinv, T f1, T f2, size_t n)
  {
    GRID_STRIDE(i, n)
    {
      if (MODE == 0)
        {
          x_old[i] = T(0);
          x[i]     = f2 * dinv[i] * b[i];
        }
      else
        {
          const T xi = x[i];
          T       xn = xi + f2 * dinv[i] * (b[i] - t[i]);
          if (MODE == 2)
            xn += f1 * (xi - x_old[i]);
          x_old[i] = xi;
          x[i]     = xn;
        }
    }
  }

  // x = f2 * dinv * b (first Chebyshev iterate from a zero start, x_old implied zero)
  template <typename T>
  __global__ void __launch_bounds__(256)
    k_cheb_init(T *__restrict__ x, const T *__restrict__ b, const T *__restrict__ dinv, T f2, size_t n)
  {
    GRID_STRIDE(i, n) x[i] = f2 * dinv[i] * b[i];
  }

  // Chebyshev update on the constrained rows, where (A x)_c = x_c (laplace_operator.h:736-737);
  // mode as in mgx_brick.hip: 2 general, 3 first step, 4 x_old == 0, 5 x = f0 D^-1 b computed
  // here and x_old == 0, 6 x_old = f0 D^-1 b computed here
  template <typename T>
  __global__ void __launch_bounds__(256)
    k_cheb_constrained(int mode, const T *__restrict__ x, T *__restrict__ out, const T *__restrict__ b,
                       const T *__restrict__ dinv, T f1, T f2, const uint32_t *__restrict__ list, uint32_t count,
                       const T *__restrict__ ax, const T *old, T f0)
  {
    GRID_STRIDE(i, count)
    {
      const uint32_t c  = list[i];
      const T        xi = mode == 5 ? f0 * dinv[c] * b[c] : x[c];
      T              xn = xi + f2 * dinv[c] * (b[c] - (ax ? ax[c] : xi));
      if (mode == 2)
        xn += f1 * (xi - old[c]);
      else if (mode == 6)
        xn += f1 * (xi - f0 * dinv[c] * b[c]);
      else if (mode == 4 || mode == 5)
        xn += f1 * xi;
      out[c] = xn;
    }
  }

  // fused residual + restriction in its scratch form: the coarse vector from the bricks' blocks of restricted values
  template <typename T>
  __global__ void __launch_bounds__(256)
    k_coarse_assemble(T *__restrict__ coarse, const T *__restrict__ scratch, const uint32_t *__restrict__ start,
                      const uint32_t *__restrict__ pos, uint32_t n)
  {
    GRID_STRIDE(d, n)
    {
      T sum = T(0);
      for (uint32_t k = start[d]; k < start[d + 1]; ++k)
        sum += scratch[pos[k]];
      coarse[d] = sum;
    }
  }

  // Fused level transfers on a decomposed mesh (TransferData::ifr_*, ifp_*): the part of the DoFs on the rank
  // interface, whose sums of A x are complete only after the exchange.  Restriction of their residuals, one thread
  // per coarse DoF (fixed order of the additions) ...
  // (LANES threads per row, the entries dealt out round-robin and the partial sums folded in a fixed tree: a thread
  // per row walks up to (2p+1)^2 dependent gathers -- measured 87 us for 48 000 rows at p = 4)
  template <typename T, int LANES>
  __global__ void __launch_bounds__(256)
    k_interface_restrict(const uint32_t *__restrict__ cdof, const uint32_t *__restrict__ start, const uint32_t *__restrict__ fdof,
                         const T *__restrict__ w, uint32_t n, T *__restrict__ coarse, const T *__restrict__ b,
                         const T *__restrict__ ax)
  {
    const uint32_t lane = threadIdx.x % LANES;
    for (uint64_t i = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) / LANES; i < n;
         i += (uint64_t)gridDim.x * blockDim.x / LANES)
      {
        T sum = T(0);
        for (uint32_t k = start[i] + lane; k < start[i + 1]; k += LANES)
          sum += w[k] * (ax ? b[fdof[k]] - ax[fdof[k]] : b[fdof[k]]);
#pragma unroll
        for (int o = LANES / 2; o > 0; o >>= 1)
          sum += __shfl_xor(sum, o, LANES);
        if (lane == 0)
          coarse[cdof[i]] += sum;
      }
  }

  // ... and the first Chebyshev iteration from x + P e (BrickMode kChebFirstProlong) on the shared DoFs
  template <typename T, int LANES>
  __global__ void __launch_bounds__(256)
    k_interface_prolong_cheb(const uint32_t *__restrict__ shared, const uint32_t *__restrict__ start,
                             const uint32_t *__restrict__ cdof, const T *__restrict__ w, uint32_t n, const T *__restrict__ e,
                             T *__restrict__ x, T *__restrict__ out, const T *__restrict__ b, const T *__restrict__ dinv, T f2,
                             const T *__restrict__ ax)
  {
    const uint32_t lane = threadIdx.x % LANES;
    for (uint64_t i = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) / LANES; i < n; i += (uint64_t)gridDim.x * blockDim.x / LANES)
      {
        const uint32_t d  = shared[i];
        T              pe = T(0);
        for (uint32_t k = start[i] + lane; k < start[i + 1]; k += LANES)
          pe += w[k] * e[cdof[k]];
#pragma unroll
        for (int o = LANES / 2; o > 0; o >>= 1)
          pe += __shfl_xor(pe, o, LANES);
        if (lane == 0)
          {
            const T xi = x[d] + pe;
            out[d]     = xi + f2 * dinv[d] * (b[d] - ax[d]);
            x[d]       = xi; // x_old of the next iteration
          }
      }
  }

  // Colour-free brick schedule (mgx_macro.hip, FREE): the DoFs on brick surfaces.  Entry i of the list:
  // DoF sdof[i], whose value of A x is the sum of the bricks' private values priv[spos[k]],
  // k in [sstart[i], sstart[i+1]), added in that (fixed) order.  Entries below n_carrier_only are shared
  // with other ranks: carrier[d] = sum and nothing else (completed after the exchange).  Otherwise the
  // post-operation of the brick loop (BrickMode, post_finish in mgx_macro.hip) on the completed value.
  template <typename T>
  __global__ void __launch_bounds__(256)
    k_surf_finish(int mode, const T *__restrict__ priv, const uint32_t *__restrict__ sdof,
                  const uint32_t *__restrict__ sstart, const uint32_t *__restrict__ spos, uint32_t first, uint32_t count,
                  uint32_t n_carrier_only, T *carrier, const T *x, T *out, const T *__restrict__ a,
                  const T *__restrict__ dinv, const T *old, T f1, T f2, T f0, const uint32_t *__restrict__ clist, uint32_t n_c)
  {
    // entries count ... count + n_c (Chebyshev forms on one rank): the constrained rows clist, where A x = x
    // (laplace_operator.h:736-737) -- the list kernel they would otherwise get is folded in here
    GRID_STRIDE(i0, count + n_c)
    {
      const bool ident = i0 >= count;
      uint32_t   d;
      T          sum = T(0);
      if (ident)
        d = clist[i0 - count];
      else
        {
          const uint32_t i = first + (uint32_t)i0;
          d                = sdof[i];
          for (uint32_t k = sstart[i]; k < sstart[i + 1]; ++k)
            sum += priv[spos[k]];
          if (i < n_carrier_only)
            {
              carrier[d] = sum;
              continue;
            }
        }
      if (mode == 0)
        out[d] = sum;
      else if (mode == 1)
        out[d] = a[d] - sum;
      else
        {
          const T bv = dinv[d], av = a[d];
          const T xi = mode == 5 ? f0 * bv * av : x[d];
          T       xn = xi + f2 * bv * (av - (ident ? xi : sum));
          if (mode == 2)
            xn += f1 * (xi - old[d]);
          else if (mode == 6)
            xn += f1 * (xi - f0 * bv * av);
          else if (mode == 4 || mode == 5)
            xn += f1 * xi;
          out[d] = xn;
        }
    }
  }

  // ---- interface exchange (domain decomposition) ----
  template <typename T>
  __global__ void __launch_bounds__(256)
    k_pack(T *__restrict__ buf, const T *__restrict__ v, const uint32_t *__restrict__ list, uint32_t count)
  {
    GRID_STRIDE(i, count) buf[i] = v[list[i]];
  }

  // all neighbours in one launch: entry e of the concatenated lists goes to send buffer seg[e]
  struct ExchangePtrs
  {
    void    *buf[32];
    uint32_t start[33];
  };
  template <typename T>
  __global__ void __launch_bounds__(256)
    k_pack_all(ExchangePtrs p, const T *__restrict__ v, const uint32_t *__restrict__ index,
               const uint8_t *__restrict__ seg, uint32_t total)
  {
    GRID_STRIDE(e, total)
    {
      const uint32_t k                 = seg[e];
      ((T *)p.buf[k])[e - p.start[k]] = v[index[e]];
    }
  }
  // one thread per interface DoF: its own partial sum and the neighbours' contributions are added
  // in ascending rank order (identical on every rank that holds the DoF => bitwise equal copies)
  template <typename T>
  __global__ void __launch_bounds__(256)
    k_unpack_ordered(ExchangePtrs p, T *__restrict__ v, const uint32_t *__restrict__ shared,
                     const uint32_t *__restrict__ csr_start, const uint8_t *__restrict__ csr_k,
                     const uint32_t *__restrict__ csr_pos, uint32_t n_shared)
  {
    GRID_STRIDE(j, n_shared)
    {
      const uint32_t dof = shared[j];
      T              sum = T(0);
      for (uint32_t c = csr_start[j]; c < csr_start[j + 1]; ++c)
        {
          const uint32_t k = csr_k[c];
          sum += k == 255u ? v[dof] : ((const T *)p.buf[k])[csr_pos[c]];
        }
      v[dof] = sum;
    }
  }

  // the same with the Chebyshev update of the fused brick forms applied to the completed sums (k_cheb_constrained
  // with ax = v), and to the constrained rows behind them (entries n_shared ... n_shared + n_c: A x = x)
  template <typename T>
  __global__ void __launch_bounds__(256)
    k_unpack_ordered_cheb(ExchangePtrs p, T *__restrict__ v, const uint32_t *__restrict__ shared,
                          const uint32_t *__restrict__ csr_start, const uint8_t *__restrict__ csr_k,
                          const uint32_t *__restrict__ csr_pos, uint32_t n_shared, int mode, const T *x, T *out,
                          const T *__restrict__ b, const T *__restrict__ dinv, const T *old, T f1, T f2, T f0,
                          const uint32_t *__restrict__ clist, uint32_t n_c)
  {
    GRID_STRIDE(j, n_shared + n_c)
    {
      const bool     ident = j >= n_shared;
      const uint32_t dof   = ident ? clist[j - n_shared] : shared[j];
      T              sum   = T(0);
      if (!ident)
        {
          for (uint32_t c = csr_start[j]; c < csr_start[j + 1]; ++c)
            {
              const uint32_t k = csr_k[c];
              sum += k == 255u ? v[dof] : ((const T *)p.buf[k])[csr_pos[c]];
            }
          v[dof] = sum;
        }
      const T xi = mode == 5 ? f0 * dinv[dof] * b[dof] : x[dof];
      T       xn = xi + f2 * dinv[dof] * (b[dof] - (ident ? xi : sum));
      if (mode == 2)
        xn += f1 * (xi - old[dof]);
      else if (mode == 6)
        xn += f1 * (xi - f0 * dinv[dof] * b[dof]);
      else if (mode == 4 || mode == 5)
        xn += f1 * xi;
      out[dof] = xn;
    }
  }

  template <typename T>
  __global__ void __launch_bounds__(256)
    k_unpack_add(T *__restrict__ v, const T *__restrict__ buf, const uint32_t *__restrict__ list, uint32_t count)
  {
    GRID_STRIDE(i, count) v[list[i]] += buf[i];
  }

  template <typename T>
  __global__ void __launch_bounds__(256)
    k_list_residual(T *__restrict__ res, const T *__restrict__ rhs, const uint32_t *__restrict__ list,
                    uint32_t count)
  {
    GRID_STRIDE(i, count)
    {
      const uint32_t c = list[i];
      res[c]           = rhs[c] - res[c];
    }
  }

  template <typename T>
  __global__ void __launch_bounds__(256)
    k_index_mod11(T *__restrict__ v, const uint32_t *__restrict__ gid, double mean, size_t n)
  {
    GRID_STRIDE(i, n) v[i] = (T)((double)((gid ? gid[i] : (uint32_t)i) % 11u) - mean);
  }

  // ---- reductions: deterministic two-stage sum (block partials, then one block) ----
  __device__ __forceinline__ double wave_sum(double v)
  {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
      v += __shfl_down(v, o, 64);
    return v;
  }

  __device__ __forceinline__ double block_sum(double v)
  {
    __shared__ double ws[4];
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0)
      ws[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = 0;
    if (threadIdx.x == 0)
      r = ws[0] + ws[1] + ws[2] + ws[3];
    __syncthreads();
    return r;
  }

  template <typename T>
  __global__ void __launch_bounds__(256)
    k_dot_partial(const T *__restrict__ x, const T *__restrict__ y, size_t n, double *__restrict__ partial)
  {
    double s = 0;
    GRID_STRIDE(i, n) s += (double)x[i] * (double)y[i];
    s = block_sum(s);
    if (threadIdx.x == 0)
      partial[blockIdx.x] = s;
  }

  __global__ void __launch_bounds__(256) k_reduce_final(const double *__restrict__ partial, int count,
                                                        double *__restrict__ result)
  {
    double s = 0;
    for (int i = threadIdx.x; i < count; i += 256)
      s += partial[i];
    s = block_sum(s);
    if (threadIdx.x == 0)
      *result = s;
  }

  template <typename T>
  __global__ void __launch_bounds__(256)
    k_cg_update(T *__restrict__ x, T *__restrict__ r, const T *__restrict__ d, const T *__restrict__ h, T alpha,
                size_t n, double *__restrict__ partial)
  {
    double s = 0;
    GRID_STRIDE(i, n)
    {
      x[i] += alpha * d[i];
      const T ri = r[i] - alpha * h[i];
      r[i]       = ri;
      s += (double)ri * (double)ri;
    }
    s = block_sum(s);
    if (threadIdx.x == 0)
      partial[blockIdx.x] = s;
  }

  template <typename T>
  __global__ void __launch_bounds__(256) k_xpby(T *__restrict__ d, const T *__restrict__ z, T beta, size_t n)
  {
    GRID_STRIDE(i, n) d[i] = z[i] + beta * d[i];
  }

  // Mixed-precision PCG (fp64 outer iteration, fp32 V-cycle: the reference's default, poisson_cube/program.cc:76-77):
  // the two precision-converting copies around the preconditioner (multigrid_solver.h:503, 507) are folded into
  // the CG kernels next to them -- the residual update writes the fp32 defect the V-cycle reads, the r.z product and
  // the direction update read the V-cycle's fp32 result.  Same values as the copies would have produced.
  __global__ void __launch_bounds__(256)
    k_cg_update_f32copy(double *__restrict__ x, double *__restrict__ r, const double *__restrict__ d,
                        const double *__restrict__ h, double alpha, size_t n, float *__restrict__ r32,
                        double *__restrict__ partial)
  {
    double s = 0;
    GRID_STRIDE(i, n)
    {
      x[i] += alpha * d[i];
      const double ri = r[i] - alpha * h[i];
      r[i]            = ri;
      r32[i]          = (float)ri;
      s += ri * ri;
    }
    s = block_sum(s);
    if (threadIdx.x == 0)
      partial[blockIdx.x] = s;
  }
  __global__ void __launch_bounds__(256)
    k_dot_partial_f64_f32(const double *__restrict__ x, const float *__restrict__ y, size_t n, double *__restrict__ partial)
  {
    double s = 0;
    GRID_STRIDE(i, n) s += x[i] * (double)y[i];
    s = block_sum(s);
    if (threadIdx.x == 0)
      partial[blockIdx.x] = s;
  }
  __global__ void __launch_bounds__(256) k_xpby_f64_f32(double *__restrict__ d, const float *__restrict__ z, double beta, size_t n)
  {
    GRID_STRIDE(i, n) d[i] = (double)z[i] + beta * d[i];
  }

  template <typename T>
  __global__ void __launch_bounds__(256)
    k_jacobi_dot(T *__restrict__ z, const T *__restrict__ dinv, const T *__restrict__ r, size_t n,
                 double *__restrict__ partial)
  {
    double s = 0;
    GRID_STRIDE(i, n)
    {
      const T zi = dinv[i] * r[i];
      z[i]       = zi;
      s += (double)r[i] * (double)zi;
    }
    s = block_sum(s);
    if (threadIdx.x == 0)
      partial[blockIdx.x] = s;
  }

  template <typename T>
  __global__ void __launch_bounds__(256)
    k_dot_list_partial(const T *__restrict__ x, const T *__restrict__ y, const uint32_t *__restrict__ list,
                       uint32_t count, double *__restrict__ partial)
  {
    double s = 0;
    GRID_STRIDE(i, count)
    {
      const uint32_t c = list[i];
      s += (double)x[c] * (double)y[c];
    }
    s = block_sum(s);
    if (threadIdx.x == 0)
      partial[blockIdx.x] = s;
  }

  static inline dim3 reduce_grid(size_t n)
  {
    size_t nb = (n + 255) / 256;
    if (nb > (size_t)kDotBlocks)
      nb = kDotBlocks;
    if (nb < 1)
      nb = 1;
    return dim3((unsigned)nb);
  }

#define BY_NUMBER(number, ...)    \
  if ((number) == 1)              \
    {                             \
      using T = double;           \
      __VA_ARGS__;                \
    }                             \
  else                            \
    {                             \
      using T = float;            \
      __VA_ARGS__;                \
    }

  void launch_copy_cast(hipStream_t s, void *dst, int dn, const void *src, int sn, size_t n)
  {
    if (n == 0)
      return;
    const dim3 g = stream_grid(n);
    if (dn == 1 && sn == 1)
      hipLaunchKernelGGL((k_copy_cast<double, double>), g, dim3(256), 0, s, (double *)dst, (const double *)src, n);
    else if (dn == 1 && sn == 0)
      hipLaunchKernelGGL((k_copy_cast<double, float>), g, dim3(256), 0, s, (double *)dst, (const float *)src, n);
    else if (dn == 0 && sn == 1)
      hipLaunchKernelGGL((k_copy_cast<float, double>), g, dim3(256), 0, s, (float *)dst, (const double *)src, n);
    else
      hipLaunchKernelGGL((k_copy_cast<float, float>), g, dim3(256), 0, s, (float *)dst, (const float *)src, n);
  }

  void launch_add_cast(hipStream_t s, void *dst, int dn, const void *src, int sn, size_t n)
  {
    if (n == 0)
      return;
    const dim3 g = stream_grid(n);
    if (dn == 1 && sn == 1)
      hipLaunchKernelGGL((k_add_cast<double, double>), g, dim3(256), 0, s, (double *)dst, (const double *)src, n);
    else if (dn == 1 && sn == 0)
      hipLaunchKernelGGL((k_add_cast<double, float>), g, dim3(256), 0, s, (double *)dst, (const float *)src, n);
    else if (dn == 0 && sn == 1)
      hipLaunchKernelGGL((k_add_cast<float, double>), g, dim3(256), 0, s, (float *)dst, (const double *)src, n);
    else
      hipLaunchKernelGGL((k_add_cast<float, float>), g, dim3(256), 0, s, (float *)dst, (const float *)src, n);
  }

  void launch_sadd(hipStream_t s, int number, void *x, double sx, double a, const void *v, size_t n)
  {
    if (n == 0)
      return;
    BY_NUMBER(number, hipLaunchKernelGGL((k_sadd<T>), stream_grid(n), dim3(256), 0, s, (T *)x, (T)sx, (T)a,
                                         (const T *)v, n));
  }

  void launch_rhs_minus(hipStream_t s, int number, void *res, const void *rhs, size_t n)
  {
    if (n == 0)
      return;
    BY_NUMBER(number,
              hipLaunchKernelGGL((k_rhs_minus<T>), stream_grid(n), dim3(256), 0, s, (T *)res, (const T *)rhs, n));
  }

  void launch_constrained_copy(hipStream_t s, int number, void *dst, const void *src, const uint32_t *list,
                               uint32_t count)
  {
    if (count == 0)
      return;
    BY_NUMBER(number, hipLaunchKernelGGL((k_constrained_copy<T>), stream_grid(count), dim3(256), 0, s, (T *)dst,
                                         (const T *)src, list, count));
  }

  void launch_constrained_residual(hipStream_t s, int number, void *res, const void *rhs, const void *lhs,
                                   const uint32_t *list, uint32_t count)
  {
    if (count == 0)
      return;
    BY_NUMBER(number, hipLaunchKernelGGL((k_constrained_residual<T>), stream_grid(count), dim3(256), 0, s,
                                         (T *)res, (const T *)rhs, (const T *)lhs, list, count));
  }

  void launch_constrained_set(hipStream_t s, int number, void *v, double value, const uint32_t *list,
                              uint32_t count)
  {
    if (count == 0)
      return;
    BY_NUMBER(number, hipLaunchKernelGGL((k_constrained_set<T>), stream_grid(count), dim3(256), 0, s, (T *)v,
                                         (T)value, list, count));
  }

  void launch_invert(hipStream_t s, int number, void *v, size_t n)
  {
    if (n == 0)
      return;
    BY_NUMBER(number, hipLaunchKernelGGL((k_invert<T>), stream_grid(n), dim3(256), 0, s, (T *)v, n));
  }

  void launch_scatter_values(hipStream_t s, int number, void *v, const uint32_t *idx_dev, const double *val_dev,
                             uint32_t count)
  {
    if (count == 0)
      return;
    BY_NUMBER(number, hipLaunchKernelGGL((k_scatter_values<T>), stream_grid(count), dim3(256), 0, s, (T *)v,
                                         idx_dev, val_dev, count));
  }

  void launch_cheb_update(hipStream_t s, int number, int mode, void *x, void *x_old, const void *b,
                          const void *t, const void *dinv, double f1, double f2, size_t n)
  {
    if (n == 0)
      return;
    const dim3 g = stream_grid(n);
    BY_NUMBER(
      number, if (mode == 0) hipLaunchKernelGGL((k_cheb_update<T, 0>), g, dim3(256), 0, s, (T *)x, (T *)x_old,
                                                (const T *)b, (const T *)t, (const T *)dinv, (T)f1, (T)f2, n);
      else if (mode == 1) hipLaunchKernelGGL((k_cheb_update<T, 1>), g, dim3(256), 0, s, (T *)x, (T *)x_old,
                                             (const T *)b, (const T *)t, (const T *)dinv, (T)f1, (T)f2, n);
      else hipLaunchKernelGGL((k_cheb_update<T, 2>), g, dim3(256), 0, s, (T *)x, (T *)x_old, (const T *)b,
                              (const T *)t, (const T *)dinv, (T)f1, (T)f2, n));
  }

  void launch_cheb_init(hipStream_t s, int number, void *x, const void *b, const void *dinv, double f2, size_t n)
  {
    if (n == 0)
      return;
    BY_NUMBER(number, hipLaunchKernelGGL((k_cheb_init<T>), stream_grid(n), dim3(256), 0, s, (T *)x, (const T *)b,
                                         (const T *)dinv, (T)f2, n));
  }

  void launch_cheb_constrained(hipStream_t s, int number, int mode, const void *x, void *out, const void *b,
                               const void *dinv, double f1, double f2, const uint32_t *list, uint32_t count,
                               const void *ax, const void *old, double f0)
  {
    if (count == 0)
      return;
    if (!old)
      old = out;
    if (!x)
      x = b; // mode 5 never reads it
    BY_NUMBER(number, hipLaunchKernelGGL((k_cheb_constrained<T>), stream_grid(count), dim3(256), 0, s, mode,
                                         (const T *)x, (T *)out, (const T *)b, (const T *)dinv, (T)f1, (T)f2, list,
                                         count, (const T *)ax, (const T *)old, (T)f0));
  }

  void launch_coarse_assemble(hipStream_t s, int number, const TransferData &tr, void *coarse)
  {
    const uint32_t n = tr.coarse->n_dofs;
    if (n == 0)
      return;
    BY_NUMBER(number, hipLaunchKernelGGL((k_coarse_assemble<T>), stream_grid(n), dim3(256), 0, s, (T *)coarse,
                                         (const T *)tr.coarse_scratch, tr.cs_start, tr.cs_pos, n));
  }

  void launch_interface_restrict(hipStream_t s, int number, const TransferData &tr, void *coarse, const void *b, const void *ax)
  {
    if (tr.n_ifr == 0)
      return;
    BY_NUMBER(number, hipLaunchKernelGGL((k_interface_restrict<T, 16>), stream_grid((size_t)tr.n_ifr * 16), dim3(256), 0, s, tr.ifr_cdof,
                                         tr.ifr_start, tr.ifr_fdof, (const T *)tr.ifr_w, tr.n_ifr, (T *)coarse, (const T *)b,
                                         (const T *)ax));
  }

  void launch_interface_prolong_cheb(hipStream_t s, int number, const TransferData &tr, const uint32_t *shared, const void *e,
                                     void *x, void *out, const void *b, const void *dinv, double f2, const void *ax)
  {
    if (tr.n_ifp == 0)
      return;
    BY_NUMBER(number, hipLaunchKernelGGL((k_interface_prolong_cheb<T, 8>), stream_grid((size_t)tr.n_ifp * 8), dim3(256), 0, s, shared,
                                         tr.ifp_start, tr.ifp_cdof, (const T *)tr.ifp_w, tr.n_ifp, (const T *)e, (T *)x,
                                         (T *)out, (const T *)b, (const T *)dinv, (T)f2, (const T *)ax));
  }

  void launch_surf_finish(hipStream_t s, const OperatorData &op, int mode, uint32_t first, uint32_t count, void *carrier,
                          const void *x, void *out, const void *a, const void *dinv, const void *old, double f1, double f2,
                          double f0, const uint32_t *constrained, uint32_t n_constrained, const FreeSchedule *schedule)
  {
    if (mode < 2)
      n_constrained = 0; // the identity rows of the plain and residual forms are the caller's
    if (count + n_constrained == 0)
      return;
    const FreeSchedule &bd = schedule ? *schedule : op.bricks.fr;
    if (!old)
      old = out;
    if (!x)
      x = a; // mode 5 never reads it
    BY_NUMBER(op.number, hipLaunchKernelGGL((k_surf_finish<T>), stream_grid((size_t)count + n_constrained), dim3(256), 0, s, mode,
                                            (const T *)bd.priv, bd.surf_dof, bd.surf_start, bd.surf_pos, first, count,
                                            bd.n_surf_shared, (T *)carrier, (const T *)x, (T *)out, (const T *)a,
                                            (const T *)dinv, (const T *)old, (T)f1, (T)f2, (T)f0, constrained, n_constrained));
  }

  // dst[i] = 0 for i < n_head, dst[i] = src[i] behind: the zeroing before a cell loop that scatters
  // with atomics and the identity on the (trailing) constrained rows in one launch
  template <typename T>
  __global__ void k_zero_head_copy_tail(T *__restrict__ dst, const T *__restrict__ src, uint32_t n_head, uint32_t n)
  {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
      dst[i] = i < n_head ? T(0) : src[i];
  }

  void launch_zero_head_copy_tail(hipStream_t s, int number, void *dst, const void *src, uint32_t n_head, uint32_t n)
  {
    if (n == 0)
      return;
    BY_NUMBER(number, hipLaunchKernelGGL((k_zero_head_copy_tail<T>), stream_grid(n), dim3(256), 0, s, (T *)dst,
                                         (const T *)src, n_head, n));
  }

  // dst[map[i]] = src[i] where mask[i] (coarse-level agglomeration: every DoF is written by its owner)
  template <typename T>
  __global__ void k_scatter_map(T *__restrict__ dst, const T *__restrict__ src, const uint32_t *__restrict__ map,
                                const uint8_t *__restrict__ mask, uint32_t n)
  {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
      if (mask[i])
        dst[map[i]] = src[i];
  }

  void launch_scatter_map(hipStream_t s, int number, void *dst, const void *src, const uint32_t *map,
                          const uint8_t *mask, uint32_t n)
  {
    if (n == 0)
      return;
    BY_NUMBER(number, hipLaunchKernelGGL((k_scatter_map<T>), stream_grid(n), dim3(256), 0, s, (T *)dst, (const T *)src,
                                         map, mask, n));
  }

  void launch_pack(hipStream_t s, int number, void *buf, const void *v, const uint32_t *list, uint32_t count)
  {
    if (count == 0)
      return;
    BY_NUMBER(number, hipLaunchKernelGGL((k_pack<T>), stream_grid(count), dim3(256), 0, s, (T *)buf, (const T *)v,
                                         list, count));
  }

  void launch_pack_all(hipStream_t s, int number, void *const *send, const uint32_t *start, int n_neighbors,
                       const void *v, const uint32_t *index, const uint8_t *seg, uint32_t total)
  {
    if (total == 0)
      return;
    ExchangePtrs p{};
    for (int k = 0; k < n_neighbors; ++k)
      {
        p.buf[k]   = send[k];
        p.start[k] = start[k];
      }
    BY_NUMBER(number, hipLaunchKernelGGL((k_pack_all<T>), stream_grid(total), dim3(256), 0, s, p, (const T *)v, index,
                                         seg, total));
  }

  void launch_unpack_ordered(hipStream_t s, int number, void *const *recv, int n_neighbors, void *v,
                             const uint32_t *shared, const uint32_t *csr_start, const uint8_t *csr_k,
                             const uint32_t *csr_pos, uint32_t n_shared)
  {
    if (n_shared == 0)
      return;
    ExchangePtrs p{};
    for (int k = 0; k < n_neighbors; ++k)
      p.buf[k] = recv[k];
    BY_NUMBER(number, hipLaunchKernelGGL((k_unpack_ordered<T>), stream_grid(n_shared), dim3(256), 0, s, p, (T *)v,
                                         shared, csr_start, csr_k, csr_pos, n_shared));
  }

  void launch_unpack_ordered_cheb(hipStream_t s, int number, void *const *recv, int n_neighbors, void *v,
                                  const uint32_t *shared, const uint32_t *csr_start, const uint8_t *csr_k,
                                  const uint32_t *csr_pos, uint32_t n_shared, const ChebList &c)
  {
    if (n_shared + c.n_constrained == 0)
      return;
    ExchangePtrs p{};
    for (int k = 0; k < n_neighbors; ++k)
      p.buf[k] = recv[k];
    const void *old = c.old ? c.old : c.out, *x = c.x ? c.x : c.b; // mode 5 never reads x
    BY_NUMBER(number, hipLaunchKernelGGL((k_unpack_ordered_cheb<T>), stream_grid((size_t)n_shared + c.n_constrained), dim3(256), 0,
                                         s, p, (T *)v, shared, csr_start, csr_k, csr_pos, n_shared, c.mode, (const T *)x,
                                         (T *)c.out, (const T *)c.b, (const T *)c.dinv, (const T *)old, (T)c.f1, (T)c.f2,
                                         (T)c.f0, c.constrained, c.n_constrained));
  }

  void launch_unpack_add(hipStream_t s, int number, void *v, const void *buf, const uint32_t *list, uint32_t count)
  {
    if (count == 0)
      return;
    BY_NUMBER(number, hipLaunchKernelGGL((k_unpack_add<T>), stream_grid(count), dim3(256), 0, s, (T *)v,
                                         (const T *)buf, list, count));
  }

  void launch_list_residual(hipStream_t s, int number, void *res, const void *rhs, const uint32_t *list,
                            uint32_t count)
  {
    if (count == 0)
      return;
    BY_NUMBER(number, hipLaunchKernelGGL((k_list_residual<T>), stream_grid(count), dim3(256), 0, s, (T *)res,
                                         (const T *)rhs, list, count));
  }

  void launch_index_mod11(hipStream_t s, int number, void *v, const uint32_t *gid, double mean, size_t n)
  {
    if (n == 0)
      return;
    BY_NUMBER(number, hipLaunchKernelGGL((k_index_mod11<T>), stream_grid(n), dim3(256), 0, s, (T *)v, gid, mean, n));
  }

  void launch_dot(hipStream_t s, int number, const void *x, const void *y, size_t n, double *partial_dev,
                  double *result_dev)
  {
    const dim3 g = reduce_grid(n);
    BY_NUMBER(number, hipLaunchKernelGGL((k_dot_partial<T>), g, dim3(256), 0, s, (const T *)x, (const T *)y, n,
                                         partial_dev));
    hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(256), 0, s, partial_dev, (int)g.x, result_dev);
  }

  void launch_dot_list(hipStream_t s, int number, const void *x, const void *y, const uint32_t *list,
                       uint32_t count, double *partial_dev, double *result_dev)
  {
    const dim3 g = reduce_grid(count);
    BY_NUMBER(number, hipLaunchKernelGGL((k_dot_list_partial<T>), g, dim3(256), 0, s, (const T *)x, (const T *)y,
                                         list, count, partial_dev));
    hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(256), 0, s, partial_dev, (int)g.x, result_dev);
  }

  void launch_cg_update(hipStream_t s, int number, void *x, void *r, const void *d, const void *h, double alpha,
                        size_t n, double *partial_dev, double *result_dev)
  {
    const dim3 g = reduce_grid(n);
    BY_NUMBER(number, hipLaunchKernelGGL((k_cg_update<T>), g, dim3(256), 0, s, (T *)x, (T *)r, (const T *)d,
                                         (const T *)h, (T)alpha, n, partial_dev));
    hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(256), 0, s, partial_dev, (int)g.x, result_dev);
  }

  void launch_cg_update_f32copy(hipStream_t s, double *x, double *r, const double *d, const double *h, double alpha, size_t n,
                                float *r32, double *partial_dev, double *result_dev)
  {
    const dim3 g = reduce_grid(n);
    hipLaunchKernelGGL(k_cg_update_f32copy, g, dim3(256), 0, s, x, r, d, h, alpha, n, r32, partial_dev);
    hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(256), 0, s, partial_dev, (int)g.x, result_dev);
  }

  void launch_dot_f64_f32(hipStream_t s, const double *x, const float *y, size_t n, double *partial_dev, double *result_dev)
  {
    const dim3 g = reduce_grid(n);
    hipLaunchKernelGGL(k_dot_partial_f64_f32, g, dim3(256), 0, s, x, y, n, partial_dev);
    hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(256), 0, s, partial_dev, (int)g.x, result_dev);
  }

  void launch_xpby_f64_f32(hipStream_t s, double *d, const float *z, double beta, size_t n)
  {
    if (n == 0)
      return;
    hipLaunchKernelGGL(k_xpby_f64_f32, stream_grid(n), dim3(256), 0, s, d, z, beta, n);
  }

  void launch_xpby(hipStream_t s, int number, void *d, const void *z, double beta, size_t n)
  {
    if (n == 0)
      return;
    BY_NUMBER(number,
              hipLaunchKernelGGL((k_xpby<T>), stream_grid(n), dim3(256), 0, s, (T *)d, (const T *)z, (T)beta, n));
  }

  void launch_jacobi_dot(hipStream_t s, int number, void *z, const void *dinv, const void *r, size_t n,
                         double *partial_dev, double *result_dev)
  {
    const dim3 g = reduce_grid(n);
    BY_NUMBER(number, hipLaunchKernelGGL((k_jacobi_dot<T>), g, dim3(256), 0, s, (T *)z, (const T *)dinv,
                                         (const T *)r, n, partial_dev));
    hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(256), 0, s, partial_dev, (int)g.x, result_dev);
  }

  // ------------------------------------------------------------------------------------------
  // fused PCG helpers (vmult_with_cg_update / vmult_with_residual_update of the reference)
  // ------------------------------------------------------------------------------------------
  // quadruple of block sums, appended to a partials array that launch_reduce4 adds up in order
  __device__ __forceinline__ void block_sum4(double (&s)[4], double *__restrict__ out)
  {
    for (int k = 0; k < 4; ++k)
      {
        const double t = block_sum(s[k]);
        if (threadIdx.x == 0)
          out[4 * blockIdx.x + k] = t;
        __syncthreads();
      }
  }

  // laplace_operator.h:655-688 on an index list (the constrained rows, which the cell loop never
  // touches): x += alpha p ; p = beta p + q ; q = 0   (alpha == 0: p = q ; q = 0);  sums: r.r only
  template <typename T>
  __global__ void __launch_bounds__(256)
    k_cg_list_update(const uint32_t *__restrict__ list, uint32_t count, T alpha, T beta, const T *__restrict__ r,
                     T *__restrict__ q, T *__restrict__ p, T *__restrict__ x, double *__restrict__ partial)
  {
    double s[4] = {0., 0., 0., 0.};
    GRID_STRIDE(j, count)
    {
      const uint32_t i = list[j];
      if (alpha == T(0))
        p[i] = q[i];
      else
        {
          x[i] += alpha * p[i];
          p[i] = beta * p[i] + q[i];
        }
      q[i] = T(0);
      s[1] += (double)r[i] * (double)r[i];
    }
    block_sum4(s, partial);
  }

  // the same over a whole vector (levels without the fused brick kernel), before the matvec
  template <typename T>
  __global__ void __launch_bounds__(256)
    k_cg_pre(T *__restrict__ x, T *__restrict__ p, T *__restrict__ q, T alpha, T beta, size_t n)
  {
    GRID_STRIDE(i, n)
    {
      if (alpha == T(0))
        p[i] = q[i];
      else
        {
          x[i] += alpha * p[i];
          p[i] = beta * p[i] + q[i];
        }
      q[i] = T(0);
    }
  }

  // q.p, r.r, q.r, q.q in one pass
  template <typename T>
  __global__ void __launch_bounds__(256)
    k_dot4(const T *__restrict__ q, const T *__restrict__ p, const T *__restrict__ r, size_t n, double *__restrict__ partial)
  {
    double s[4] = {0., 0., 0., 0.};
    GRID_STRIDE(i, n)
    {
      const double qi = q[i], pi = p[i], ri = r[i];
      s[0] += qi * pi;
      s[1] += ri * ri;
      s[2] += qi * ri;
      s[3] += qi * qi;
    }
    block_sum4(s, partial);
  }

  // multigrid_solver.h:527-534: defect = residual + factor update (precision cast)
  template <typename TD>
  __global__ void __launch_bounds__(256)
    k_residual_pre(TD *__restrict__ defect, const double *__restrict__ residual, const double *__restrict__ update,
                   double factor, size_t n)
  {
    GRID_STRIDE(i, n) defect[i] = (TD)(factor != 0. ? residual[i] + factor * update[i] : residual[i]);
  }

  // multigrid_solver.h:545-603: residual += factor update ; {z.res, z.(factor update), res.res} ;
  // update = z  -- rows [n_free, n) are constrained: identity on the diagonal, z := res
  template <typename TZ>
  __global__ void __launch_bounds__(256)
    k_residual_post(const TZ *__restrict__ z, double *__restrict__ residual, double *__restrict__ update, double factor,
                    size_t n_free, size_t n, double *__restrict__ partial)
  {
    double s[4] = {0., 0., 0., 0.};
    GRID_STRIDE(i, n)
    {
      const double upd = factor != 0. ? update[i] * factor : 0.;
      const double res = residual[i] + upd;
      const double zi  = i < n_free ? (double)z[i] : res;
      residual[i]      = res;
      update[i]        = zi;
      s[0] += zi * res;
      s[1] += factor != 0. ? zi * upd : zi * res;
      s[2] += res * res;
    }
    block_sum4(s, partial);
  }

  // r += factor q ; partial quadruples {0, 0, r.r, 0} (slot 2, as launch_residual_post)
  template <typename T>
  __global__ void __launch_bounds__(256)
    k_axpy_norm(T *__restrict__ r, const T *__restrict__ q, T factor, size_t n, double *__restrict__ partial)
  {
    double s[4] = {0., 0., 0., 0.};
    GRID_STRIDE(i, n)
    {
      const T ri = r[i] + factor * q[i];
      r[i]       = ri;
      s[2] += (double)ri * (double)ri;
    }
    block_sum4(s, partial);
  }

  uint32_t launch_axpy_norm(hipStream_t s, int number, void *r, const void *q, double factor, size_t n, double *partials)
  {
    const dim3 g = reduce_grid(n);
    BY_NUMBER(number, hipLaunchKernelGGL((k_axpy_norm<T>), g, dim3(256), 0, s, (T *)r, (const T *)q, (T)factor, n, partials));
    return g.x;
  }

  uint32_t launch_cg_list_update(hipStream_t s, int number, const uint32_t *list, uint32_t count, double alpha,
                                 double beta, const void *r, void *q, void *p, void *x, double *partials)
  {
    if (count == 0)
      return 0;
    const dim3 g = reduce_grid(count);
    BY_NUMBER(number, hipLaunchKernelGGL((k_cg_list_update<T>), g, dim3(256), 0, s, list, count, (T)alpha, (T)beta,
                                         (const T *)r, (T *)q, (T *)p, (T *)x, partials));
    return g.x;
  }

  void launch_cg_pre(hipStream_t s, int number, void *x, void *p, void *q, double alpha, double beta, size_t n)
  {
    if (n == 0)
      return;
    BY_NUMBER(number, hipLaunchKernelGGL((k_cg_pre<T>), stream_grid(n), dim3(256), 0, s, (T *)x, (T *)p, (T *)q,
                                         (T)alpha, (T)beta, n));
  }

  uint32_t launch_dot4(hipStream_t s, int number, const void *q, const void *p, const void *r, size_t n, double *partials)
  {
    const dim3 g = reduce_grid(n);
    BY_NUMBER(number, hipLaunchKernelGGL((k_dot4<T>), g, dim3(256), 0, s, (const T *)q, (const T *)p, (const T *)r, n,
                                         partials));
    return g.x;
  }

  void launch_residual_pre(hipStream_t s, int defect_number, void *defect, const double *residual, const double *update,
                           double factor, size_t n)
  {
    if (n == 0)
      return;
    BY_NUMBER(defect_number, hipLaunchKernelGGL((k_residual_pre<T>), stream_grid(n), dim3(256), 0, s, (T *)defect, residual,
                                                update, factor, n));
  }

  uint32_t launch_residual_post(hipStream_t s, int z_number, const void *z, double *residual, double *update, double factor,
                                size_t n_free, size_t n, double *partials)
  {
    const dim3 g = reduce_grid(n);
    BY_NUMBER(z_number, hipLaunchKernelGGL((k_residual_post<T>), g, dim3(256), 0, s, (const T *)z, residual, update, factor,
                                           n_free, n, partials));
    return g.x;
  }
} // namespace mgx
