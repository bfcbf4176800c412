// mgx_kernels.hip -- first-generation gfx950 kernels: the per-cell Laplace cell loop with atomic
// scatter (used on levels too small for the brick schedule of mgx_brick.hip and for cell orders
// that do not form bricks), the diagonal, and the one-workgroup-per-parent transfers (fallback of
// the pipelined kernels in mgx_transfer.hip for levels with 2^29 DoFs or more).
//
// Mapping (CDNA4, 64-wide waves): a cell of FE_Q(p) has n^3 = (p+1)^3 points.  A tile of
// n x n threads owns one cell; each thread holds ONE 1D line of n values in registers, so every
// sum-factorisation sweep is an in-register (n x n)(n) product whose matrix
// entries arrive through wave-uniform scalar loads.  Between sweeps the cell is transposed
// through LDS (x-lines -> y-lines -> z-lines), one write + one read of n values per thread,
// instead of the n^2 LDS reads per sweep of a "column in registers, rows in LDS" scheme.
// Several cells are packed per 256-thread workgroup so the waves are full
// (p=4: 10 cells = 250 threads).
//
// The DoF gather/scatter uses the reference's 27-entry compressed index table
// (common/vector_access_reduced.h:11-505, restated in SURVEY.md Appendix A): thread (j,k) loads
// the x-line {left vertex/edge/face entry, p-1 contiguous interior entries, right entry}.
#include "mgx_macro_device.hpp" // buffer-descriptor access (brick_general_kernel)
#include "mgx_internal.hpp"

#include <hip/hip_runtime.h>

namespace mgx
{
#ifndef MGX_GENERAL_WG_THREADS
#define MGX_GENERAL_WG_THREADS 256 // 128 measured: no gain
#endif
  template <int P, int WG = 256>
  struct Cfg
  {
    static constexpr int N        = P + 1;
    static constexpr int LN       = N | 1; // x-line pitch, odd => conflict-free ds_read_b64
    static constexpr int TPC      = N * N; // threads per cell
    static constexpr int CPB      = (WG / TPC) < 1 ? 1 : (WG / TPC);
    static constexpr int THREADS  = ((CPB * TPC + 63) / 64) * 64;
    static constexpr int CELL_LDS = N * N * LN;
  };

  // out[a] = sum_b M[a*N+b] in[b]
  template <int N, typename T>
  __device__ __forceinline__ void mv(const T *__restrict__ M, const T (&in)[N], T (&out)[N])
  {
#pragma unroll
    for (int a = 0; a < N; ++a)
      {
        T s = M[a * N] * in[0];
#pragma unroll
        for (int b = 1; b < N; ++b)
          s = fma(M[a * N + b], in[b], s);
        out[a] = s;
      }
  }

  // out[a] = sum_b M[b*N+a] in[b]
  template <int N, typename T>
  __device__ __forceinline__ void mvT(const T *__restrict__ M, const T (&in)[N], T (&out)[N])
  {
#pragma unroll
    for (int a = 0; a < N; ++a)
      {
        T s = M[a] * in[0];
#pragma unroll
        for (int b = 1; b < N; ++b)
          s = fma(M[b * N + a], in[b], s);
        out[a] = s;
      }
  }

  // entity code (0 = low vertex plane, 1 = interior, 2 = high) and offset inside the entity of
  // the 1D node index j (vector_access_reduced.h:232-247)
  template <int P>
  __device__ __forceinline__ void node_code(int j, int &code, int &offs)
  {
    code = (j == 0) ? 0 : (j == P ? 2 : 1);
    offs = (code == 1) ? j - 1 : 0;
  }

  template <int P>
  struct LineIndex
  {
    uint32_t b0, b1, b2; // first DoF of the left / interior / right entity of this x-line
    uint32_t off;        // offset of the line inside those entities
  };

  // address computation of read_dof_values_compressed for the x-line (j,k) of `cell`
  // (vector_access_reduced.h:153-229)
  template <int P>
  __device__ __forceinline__ LineIndex<P> line_index(const uint32_t *__restrict__ idx27, uint32_t cell, int j,
                                                     int k)
  {
    int cy, oy, cz, oz;
    node_code<P>(j, cy, oy);
    node_code<P>(k, cz, oz);
    LineIndex<P>    L;
    const uint32_t *ind = idx27 + 27u * (size_t)cell + 3 * (3 * cz + cy);
    L.b0                = ind[0];
    L.b1                = ind[1];
    L.b2                = ind[2];
    L.off               = (uint32_t)((cy == 1 ? P - 1 : 1) * oz + oy);
    return L;
  }

  template <int P, typename T>
  __device__ __forceinline__ void gather_line(const T *__restrict__ src, const LineIndex<P> &L, T (&r)[P + 1])
  {
    r[0] = L.b0 != kInvalid ? src[L.b0 + L.off] : T(0);
#pragma unroll
    for (int i = 0; i < P - 1; ++i)
      r[1 + i] = L.b1 != kInvalid ? src[L.b1 + L.off * (uint32_t)(P - 1) + (uint32_t)i] : T(0);
    r[P] = L.b2 != kInvalid ? src[L.b2 + L.off] : T(0);
  }

  template <int P, typename T>
  __device__ __forceinline__ void scatter_add_line(T *__restrict__ dst, const LineIndex<P> &L,
                                                   const T (&r)[P + 1])
  {
    if (L.b0 != kInvalid)
      unsafeAtomicAdd(&dst[L.b0 + L.off], r[0]);
    if (L.b1 != kInvalid)
      {
#pragma unroll
        for (int i = 0; i < P - 1; ++i)
          unsafeAtomicAdd(&dst[L.b1 + L.off * (uint32_t)(P - 1) + (uint32_t)i], r[1 + i]);
      }
    if (L.b2 != kInvalid)
      unsafeAtomicAdd(&dst[L.b2 + L.off], r[P]);
  }

  // the same without atomics: for launches over cells of one colour (no two of them share a DoF)
  template <int P, typename T>
  __device__ __forceinline__ void scatter_add_line_plain(T *__restrict__ dst, const LineIndex<P> &L,
                                                         const T (&r)[P + 1])
  {
    if (L.b0 != kInvalid)
      dst[L.b0 + L.off] += r[0];
    if (L.b1 != kInvalid)
      {
#pragma unroll
        for (int i = 0; i < P - 1; ++i)
          dst[L.b1 + L.off * (uint32_t)(P - 1) + (uint32_t)i] += r[1 + i];
      }
    if (L.b2 != kInvalid)
      dst[L.b2 + L.off] += r[P];
  }

  // Ordered assembly (levels without a brick schedule): instead of adding into the vector, a cell
  // stores its (p+1)^3 local results at scratch[cell (p+1)^3 + (k n + j) n + i]; assemble_kernel
  // below then adds, for every DoF, its contributions in ascending cell order -- no atomics, the
  // sum does not depend on the order in which the workgroups happen to run
  template <int P, typename T>
  __device__ __forceinline__ void store_line_local(T *__restrict__ scratch, uint32_t cell, int j, int k,
                                                   const T (&r)[P + 1])
  {
    constexpr int N = P + 1;
    T            *o = scratch + (size_t)cell * (N * N * N) + (size_t)((k * N + j) * N);
#pragma unroll
    for (int i = 0; i < N; ++i)
      o[i] = r[i];
  }

  // mode 0: dst[d] = sum of the contributions of DoF d (DoFs without contributions, i.e. the
  //         constrained ones, get 0 -- or tail_src[d] for d >= n_head: the identity rows of
  //         LaplaceOperator::vmult, laplace_operator.h:592-593, when they are the tail of the vector)
  // mode 1: dst[d] += sum (DoFs without contributions untouched)
  template <typename T, int MODE>
  __global__ void __launch_bounds__(256)
    assemble_kernel(T *__restrict__ dst, const T *__restrict__ scratch, const uint32_t *__restrict__ start,
                    const uint32_t *__restrict__ pos, uint32_t n_dofs, const T *__restrict__ tail_src, uint32_t n_head)
  {
    const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= n_dofs)
      return;
    const uint32_t b = start[d], e = start[d + 1];
    T              sum = T(0);
    for (uint32_t k = b; k < e; ++k)
      sum += scratch[pos[k]];
    if (MODE == 1)
      {
        if (e > b)
          dst[d] += sum;
      }
    else
      dst[d] = (tail_src && d >= n_head) ? tail_src[d] : sum;
  }

  // the assembly with the Chebyshev update as its post-operation (ChebPost): t = (A x)[d] never reaches memory
  template <typename T, bool THREE>
  __global__ void __launch_bounds__(256)
    assemble_cheb_kernel(T *__restrict__ x, const T *__restrict__ scratch, const uint32_t *__restrict__ start,
                         const uint32_t *__restrict__ pos, uint32_t n_dofs, uint32_t n_head, T *__restrict__ x_old,
                         const T *__restrict__ b, const T *__restrict__ dinv, T f1, T f2)
  {
    const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= n_dofs)
      return;
    const T xi = x[d];
    T       t  = xi; // identity rows
    if (d < n_head)
      {
        t = T(0);
        for (uint32_t k = start[d]; k < start[d + 1]; ++k)
          t += scratch[pos[k]];
      }
    T xn = xi + f2 * dinv[d] * (b[d] - t);
    if (THREE)
      xn += f1 * (xi - x_old[d]);
    x_old[d] = xi;
    x[d]     = xn;
  }

  // ------------------------------------------------------------------------------------------
  // Cell loop: dst += sum_cells  S^T [ sum_d D_d^T (c_d w) D_d ] S  src   (diagonal coefficient)
  // = local_apply of laplace_operator.h:527-558 with the quadrature-point operation :471-487.
  // ------------------------------------------------------------------------------------------
  template <int P, typename T>
  __global__ void __launch_bounds__(Cfg<P>::THREADS)
    cell_loop_kernel(T *__restrict__ dst, const T *__restrict__ src, const uint32_t *__restrict__ idx27,
                     uint32_t n_cells, const Basis1D<T> *__restrict__ B, T c0, T c1, T c2, T *__restrict__ scratch)
  {
    using C           = Cfg<P>;
    constexpr int N   = C::N;
    constexpr int LN  = C::LN;
    constexpr int PL  = N * LN; // plane pitch
    __shared__ T U[C::CPB * C::CELL_LDS];
    __shared__ T V[C::CPB * C::CELL_LDS];

    const int      tid    = threadIdx.x;
    const int      lc     = tid / C::TPC;
    const int      t      = tid - lc * C::TPC;
    const int      a      = t % N;
    const int      b      = t / N;
    const uint32_t cell   = blockIdx.x * C::CPB + lc;
    const bool     active = (lc < C::CPB) && (cell < n_cells);
    T             *Uc     = U + (lc < C::CPB ? lc : 0) * C::CELL_LDS;
    T             *Vc     = V + (lc < C::CPB ? lc : 0) * C::CELL_LDS;

    const int xl = (b * N + a) * LN; // x-line (j=a,k=b): + i
    const int yl = b * PL + a;       // y-line (i=a,k=b): + j*LN
    const int zl = b * LN + a;       // z-line (i=a,j=b): + k*PL

    const T wa = B->w[a], wb = B->w[b];
    T       r[N], q[N], vz[N];
    LineIndex<P> L;

    // 1. gather x-lines, interpolate to quadrature points along x
    if (active)
      {
        L = line_index<P>(idx27, cell, a, b);
        gather_line<P, T>(src, L, r);
        mv<N, T>(B->S, r, q);
#pragma unroll
        for (int i = 0; i < N; ++i)
          Uc[xl + i] = q[i];
      }
    __syncthreads();
    // 2. along y
    if (active)
      {
#pragma unroll
        for (int i = 0; i < N; ++i)
          r[i] = Uc[yl + i * LN];
        mv<N, T>(B->S, r, q);
#pragma unroll
        for (int i = 0; i < N; ++i)
          Uc[yl + i * LN] = q[i];
      }
    __syncthreads();
    // 3. along z -> values at quadrature points; z-derivative pair in registers
    if (active)
      {
#pragma unroll
        for (int i = 0; i < N; ++i)
          r[i] = Uc[zl + i * PL];
        mv<N, T>(B->S, r, q); // q = u at quadrature points of this z-line
#pragma unroll
        for (int i = 0; i < N; ++i)
          Uc[zl + i * PL] = q[i];
        mv<N, T>(B->D, q, r);
        const T f = c2 * wa * wb;
#pragma unroll
        for (int i = 0; i < N; ++i)
          r[i] *= f * B->w[i];
        mvT<N, T>(B->D, r, vz);
      }
    __syncthreads();
    // 4. x-derivative pair
    if (active)
      {
#pragma unroll
        for (int i = 0; i < N; ++i)
          q[i] = Uc[xl + i];
        mv<N, T>(B->D, q, r);
        const T f = c0 * wa * wb;
#pragma unroll
        for (int i = 0; i < N; ++i)
          r[i] *= f * B->w[i];
        mvT<N, T>(B->D, r, q);
#pragma unroll
        for (int i = 0; i < N; ++i)
          Vc[xl + i] = q[i];
      }
    __syncthreads();
    // 5. y-derivative pair, accumulated
    if (active)
      {
#pragma unroll
        for (int i = 0; i < N; ++i)
          q[i] = Uc[yl + i * LN];
        mv<N, T>(B->D, q, r);
        const T f = c1 * wa * wb;
#pragma unroll
        for (int i = 0; i < N; ++i)
          r[i] *= f * B->w[i];
        mvT<N, T>(B->D, r, q);
#pragma unroll
        for (int i = 0; i < N; ++i)
          Vc[yl + i * LN] += q[i];
      }
    __syncthreads();
    // 6. add the z part, integrate along z
    if (active)
      {
#pragma unroll
        for (int i = 0; i < N; ++i)
          r[i] = Vc[zl + i * PL] + vz[i];
        mvT<N, T>(B->S, r, q);
#pragma unroll
        for (int i = 0; i < N; ++i)
          Vc[zl + i * PL] = q[i];
      }
    __syncthreads();
    // 7. along y
    if (active)
      {
#pragma unroll
        for (int i = 0; i < N; ++i)
          r[i] = Vc[yl + i * LN];
        mvT<N, T>(B->S, r, q);
#pragma unroll
        for (int i = 0; i < N; ++i)
          Vc[yl + i * LN] = q[i];
      }
    __syncthreads();
    // 8. along x, scatter-add (distribute_local_to_global_compressed,
    //    vector_access_reduced.h:255-505)
    if (active)
      {
#pragma unroll
        for (int i = 0; i < N; ++i)
          r[i] = Vc[xl + i];
        mvT<N, T>(B->S, r, q);
        if (scratch)
          store_line_local<P, T>(scratch, cell, a, b, q);
        else
          scatter_add_line<P, T>(dst, L, q);
      }
  }

  // ------------------------------------------------------------------------------------------
  // General quadrature-point operation (do_quadrature_point_operation, laplace_operator.h:436-523):
  // the full symmetric tensor, either one per mesh times the quadrature weight (affine branch
  // :447-491, PERQ = false) or one per cell and quadrature point with the weight folded in
  // (:493-522, PERQ = true; evaluate_coefficient :388-430 -- variable coefficient, curved cells).
  // All three gradient components are needed at a point at once: the x- and y-derivative arrays
  // go through LDS to the thread that owns the z-line through the point, which holds the
  // z-derivative in registers, applies the tensor and hands the x / y parts back the same way.
  // coef_q is component-major per cell, [cell][6][n^3]: the N^2 threads of a cell read every
  // component with unit stride.  109.75 B per DoF at p = 4 against 16 B of the vectors:
  // this kernel is bound by the coefficient stream.
  // ------------------------------------------------------------------------------------------
  // cell_list != nullptr: the launch covers the n_cells cells cell_list[0 .. n_cells) of one colour
  // (no shared DoFs among them) and adds to dst without atomics
  // RESID (LaplaceOperator::compute_residual, laplace_operator.h:804-845): the source is read through the
  // unconstrained index table idx_gather (boundary values in the constrained entries) and negated (:823-824),
  // rhs_q[cell][q] = f(x_q) JxW_q is added to the integrand of the test function values (:839); the rows of
  // constrained DoFs are still skipped on the way out.
  template <int P, typename T, bool PERQ, bool RESID = false>
  __global__ void __launch_bounds__((Cfg<P, MGX_GENERAL_WG_THREADS>::THREADS))
    cell_loop_general_kernel(T *__restrict__ dst, const T *__restrict__ src, const uint32_t *__restrict__ idx27,
                             uint32_t n_cells, const Basis1D<T> *__restrict__ B, const T *__restrict__ coef_q, T c0,
                             T c1, T c2, T c3, T c4, T c5, const uint32_t *__restrict__ cell_list, T *__restrict__ scratch,
                             const uint32_t *__restrict__ idx_gather = nullptr, const T *__restrict__ rhs_q = nullptr)
  {
    using C           = Cfg<P, MGX_GENERAL_WG_THREADS>;
    constexpr int N   = C::N;
    constexpr int LN  = C::LN;
    constexpr int PL  = N * LN;
    constexpr int N3  = N * N * N;
    __shared__ T U[C::CPB * C::CELL_LDS];  // values at the quadrature points, then the result
    __shared__ T GX[C::CPB * C::CELL_LDS]; // x-derivative / x part of the flux
    __shared__ T GY[C::CPB * C::CELL_LDS];

    const int      tid    = threadIdx.x;
    const int      lc     = tid / C::TPC;
    const int      t      = tid - lc * C::TPC;
    const int      a      = t % N;
    const int      b      = t / N;
    const uint32_t pos    = blockIdx.x * C::CPB + lc;
    const bool     active = (lc < C::CPB) && (pos < n_cells);
    const uint32_t cell   = (cell_list && active) ? cell_list[pos] : pos;
    const int      slot   = lc < C::CPB ? lc : 0;
    T             *Uc = U + slot * C::CELL_LDS, *Xc = GX + slot * C::CELL_LDS, *Yc = GY + slot * C::CELL_LDS;
    const int      xl = (b * N + a) * LN, yl = b * PL + a, zl = b * LN + a;
    T              r[N], q[N], gz[N];
    LineIndex<P>   L;
    if (active) // nodal -> quadrature along x
      {
        L = line_index<P>(idx27, cell, a, b);
        if (RESID)
          {
            gather_line<P, T>(src, line_index<P>(idx_gather, cell, a, b), r);
#pragma unroll
            for (int i = 0; i < N; ++i)
              r[i] = -r[i];
          }
        else
          gather_line<P, T>(src, L, r);
        mv<N, T>(B->S, r, q);
#pragma unroll
        for (int i = 0; i < N; ++i)
          Uc[xl + i] = q[i];
      }
    __syncthreads();
    if (active) // along y
      {
#pragma unroll
        for (int i = 0; i < N; ++i)
          r[i] = Uc[yl + i * LN];
        mv<N, T>(B->S, r, q);
#pragma unroll
        for (int i = 0; i < N; ++i)
          Uc[yl + i * LN] = q[i];
      }
    __syncthreads();
    if (active) // along z; z-derivative of this z-line in registers
      {
#pragma unroll
        for (int i = 0; i < N; ++i)
          r[i] = Uc[zl + i * PL];
        mv<N, T>(B->S, r, q);
#pragma unroll
        for (int i = 0; i < N; ++i)
          Uc[zl + i * PL] = q[i];
        mv<N, T>(B->D, q, gz);
      }
    __syncthreads();
    if (active) // x- and y-derivatives, to the z-line owners through LDS
      {
#pragma unroll
        for (int i = 0; i < N; ++i)
          q[i] = Uc[xl + i];
        mv<N, T>(B->D, q, r);
#pragma unroll
        for (int i = 0; i < N; ++i)
          Xc[xl + i] = r[i];
#pragma unroll
        for (int i = 0; i < N; ++i)
          q[i] = Uc[yl + i * LN];
        mv<N, T>(B->D, q, r);
#pragma unroll
        for (int i = 0; i < N; ++i)
          Yc[yl + i * LN] = r[i];
      }
    __syncthreads();
    if (active) // the tensor at the N points of this z-line: (i, j, k) = (a, b, k)
      {
        const T *cq = coef_q + (size_t)cell * 6 * N3 + (size_t)(b * N + a);
        const T  wab = B->w[a] * B->w[b];
#pragma unroll
        for (int k = 0; k < N; ++k)
          {
            T t0, t1, t2, t3, t4, t5;
            if (PERQ)
              {
                // (requesting the 6 N values of the z-line at the top of the kernel, under the gather and the
                // interpolation sweeps, was measured slower: 0.79 vs 0.55 ms on the 12.7 M DoF shell -- the loads
                // had to be volatile to stay there, which takes them past the caches)
                const T *cp = cq + k * N * N;
                t0 = cp[0], t1 = cp[N3], t2 = cp[2 * N3], t3 = cp[3 * N3], t4 = cp[4 * N3], t5 = cp[5 * N3];
              }
            else
              {
                const T w = wab * B->w[k]; // :456-457
                t0 = c0 * w, t1 = c1 * w, t2 = c2 * w, t3 = c3 * w, t4 = c4 * w, t5 = c5 * w;
              }
            const T gx = Xc[zl + k * PL], gy = Yc[zl + k * PL], g = gz[k];
            Xc[zl + k * PL] = t0 * gx + t3 * gy + t4 * g; // :473-486 / :505-518
            Yc[zl + k * PL] = t3 * gx + t1 * gy + t5 * g;
            gz[k]           = t4 * gx + t5 * gy + t2 * g;
          }
        mvT<N, T>(B->D, gz, q); // integrate the z part along z
#pragma unroll
        for (int i = 0; i < N; ++i)
          gz[i] = q[i];
      }
    __syncthreads();
    if (active) // transposed x-derivative
      {
#pragma unroll
        for (int i = 0; i < N; ++i)
          r[i] = Xc[xl + i];
        mvT<N, T>(B->D, r, q);
#pragma unroll
        for (int i = 0; i < N; ++i)
          Uc[xl + i] = q[i];
      }
    __syncthreads();
    if (active) // transposed y-derivative, accumulated
      {
#pragma unroll
        for (int i = 0; i < N; ++i)
          r[i] = Yc[yl + i * LN];
        mvT<N, T>(B->D, r, q);
#pragma unroll
        for (int i = 0; i < N; ++i)
          Uc[yl + i * LN] += q[i];
      }
    __syncthreads();
    if (active) // add the z part, quadrature -> nodal along z
      {
#pragma unroll
        for (int i = 0; i < N; ++i)
          r[i] = Uc[zl + i * PL] + gz[i];
        if (RESID && rhs_q)
          {
#pragma unroll
            for (int i = 0; i < N; ++i)
              r[i] += rhs_q[(size_t)cell * N3 + (size_t)((i * N + b) * N + a)];
          }
        mvT<N, T>(B->S, r, q);
#pragma unroll
        for (int i = 0; i < N; ++i)
          Uc[zl + i * PL] = q[i];
      }
    __syncthreads();
    if (active) // along y
      {
#pragma unroll
        for (int i = 0; i < N; ++i)
          r[i] = Uc[yl + i * LN];
        mvT<N, T>(B->S, r, q);
#pragma unroll
        for (int i = 0; i < N; ++i)
          Uc[yl + i * LN] = q[i];
      }
    __syncthreads();
    if (active) // along x, scatter-add
      {
#pragma unroll
        for (int i = 0; i < N; ++i)
          r[i] = Uc[xl + i];
        mvT<N, T>(B->S, r, q);
        if (scratch)
          store_line_local<P, T>(scratch, cell, a, b, q);
        else if (cell_list)
          scatter_add_line_plain<P, T>(dst, L, q);
        else
          scatter_add_line<P, T>(dst, L, q);
      }
  }

  // ------------------------------------------------------------------------------------------
  // Brick form of the general quadrature-point operation (round 4; vmult only, p <= 4, one rank).  The kernel above
  // hands the (p+1)^3 results of every cell to the ordered assembly through a scratch array: 2 x 15.6 B per DoF at
  // p = 4 next to the 94 B of the coefficient stream.  Here a workgroup takes a brick of 4 x 4 x 4 cells (the brick
  // tables of mgx_bricks.cpp, built from the index tables of any mesh whose cells come in bricks: the hyper_shell
  // meshes do) and adds the cells' results up in an LDS array of the brick's (4p+1)^3 points: eight rounds of the
  // eight cells of one parity class (2 i + r_x, 2 j + r_y, 2 k + r_z) -- they share no point, plain adds, fixed order.
  // Write-out as in the macro-element kernel on its one-launch schedule (mgx_macro.hip, FREE with one class): item
  // by item in the order of the item table, interior entities straight to the vector, entities on the brick surface
  // to the brick's private block; k_surf_finish adds the blocks up per DoF in a fixed order.  Cell 64 b + r + 8 s is
  // the cell of parity r in slot s = (i, j, k) of brick b = order[position in the schedule] (Morton order inside the
  // brick, verified by build_bricks).
  // ------------------------------------------------------------------------------------------
  template <int P, typename T, bool PERQ>
  __global__ void __launch_bounds__(256)
    brick_general_kernel(T *__restrict__ dst, const T *__restrict__ src, const uint32_t *__restrict__ idx27,
                         const uint32_t *__restrict__ ent, const uint32_t *__restrict__ surf_off,
                         const uint32_t *__restrict__ item_map, const uint32_t *__restrict__ order, uint32_t n_bricks,
                         const Basis1D<T> *__restrict__ B,
                         const T *__restrict__ coef_q, T c0, T c1, T c2, T c3, T c4, T c5, T *__restrict__ priv, uint32_t n_surf, uint32_t vec_bytes)
  {
    constexpr int N = P + 1, LN = N | 1, PL = N * LN, N3 = N * N * N, CELL = N * N * LN;
    constexpr int NB = 4, G = NB * P + 1, NPTS = G * G * G, NE = 729, SLOTS = 8, TPC = N * N, NT = 256;
    static_assert(SLOTS * TPC <= NT, "eight cells of n^2 threads per round");
    __shared__ T        W[NPTS];
    __shared__ T        U[SLOTS * CELL], GX[SLOTS * CELL], GY[SLOTS * CELL];
    __shared__ uint32_t E[NE], S[NE];
    const int      tid   = threadIdx.x;
    const uint32_t brick = blockIdx.x;
    if (brick >= n_bricks)
      return;
    for (int i = tid; i < NPTS; i += NT)
      W[i] = T(0);
    for (int i = tid; i < NE; i += NT)
      {
        E[i] = ent[(size_t)brick * NE + i];
        S[i] = surf_off[i];
      }
    const int  slot = tid / TPC, t = tid - slot * TPC, a = t % N, b = t / N;
    const bool active = slot < SLOTS;
    const int  sl = active ? slot : 0;
    T         *Uc = U + sl * CELL, *Xc = GX + sl * CELL, *Yc = GY + sl * CELL;
    const int  xl = (b * N + a) * LN, yl = b * PL + a, zl = b * LN + a;
    // (the tables are in the order of the brick schedule -- launch groups of build_bricks --, the cells in mesh order)
    const uint32_t cell0 = 64u * order[brick];
    // Two workgroups of four waves per CU hide no latency by themselves: every round's operands are requested a round
    // ahead through buffer descriptors (the compiler leaves such loads where they are written): the three index words
    // of the thread's x-line two rounds ahead, the five source values and the 6 n coefficients of its z-line one round ahead
    // (constrained entity: offset out of range, reads zero; two sets of coefficient registers in turn).
    const rsrc_t rs_src = make_rsrc(src, vec_bytes);
    const rsrc_t rs_idx = make_rsrc(idx27 + 27u * (size_t)cell0, 64u * 27u * 4u);
    const rsrc_t rs_cq  = make_rsrc(PERQ ? coef_q + (size_t)cell0 * 6 * N3 : coef_q, PERQ ? 64u * 6u * (uint32_t)N3 * (uint32_t)sizeof(T) : 0u);
    int cyc, oy, czc, oz;
    node_code<P>(a, cyc, oy);
    node_code<P>(b, czc, oz);
    const uint32_t loff = (uint32_t)((cyc == 1 ? P - 1 : 1) * oz + oy);       // line_index(): offset of the line in its entities
    const uint32_t iw0  = (uint32_t)(3 * (3 * czc + cyc)) * 4u;               // byte offset of its three index words in a cell's 27
    auto idx_issue = [&](int r, uint32_t(&w)[3]) {
      const uint32_t o = active && r < 8 ? (uint32_t)(r + 8 * sl) * 27u * 4u + iw0 : kOob;
#pragma unroll
      for (int k = 0; k < 3; ++k)
        w[k] = __builtin_amdgcn_raw_buffer_load_b32(rs_idx, o == kOob ? kOob : o + 4u * k, 0, 0);
    };
    auto src_issue = [&](const uint32_t(&w)[3], T(&v)[N]) {
      v[0] = buf_ld(rs_src, w[0] != kInvalid ? (w[0] + loff) * (uint32_t)sizeof(T) : kOob, T());
#pragma unroll
      for (int i = 0; i < P - 1; ++i)
        v[1 + i] = buf_ld(rs_src, w[1] != kInvalid ? (w[1] + loff * (uint32_t)(P - 1) + (uint32_t)i) * (uint32_t)sizeof(T) : kOob, T());
      v[P] = buf_ld(rs_src, w[2] != kInvalid ? (w[2] + loff) * (uint32_t)sizeof(T) : kOob, T());
    };
    auto coef_issue = [&](int r, T(&c)[PERQ ? 6 * N : 1]) {
#ifdef MGX_GB_NOCOEF // diagnostic build (wrong results): no coefficient stream
      for (int k = 0; k < (PERQ ? 6 * N : 1); ++k)
        c[k] = T(1);
      return;
#endif
      if (PERQ)
        {
          const uint32_t o = active && r < 8 ? ((uint32_t)(r + 8 * sl) * 6u * (uint32_t)N3 + (uint32_t)(b * N + a)) * (uint32_t)sizeof(T) : kOob;
#pragma unroll
          for (int k = 0; k < N; ++k)
#pragma unroll
            for (int e = 0; e < 6; ++e)
              c[PERQ ? 6 * k + e : 0] = buf_ld(rs_cq, o == kOob ? kOob : o + (uint32_t)((e * N3 + k * N * N) * (int)sizeof(T)), T());
        }
    };
    uint32_t iw[3], iwn[3];
    T        gv[N], cfa[PERQ ? 6 * N : 1], cfb[PERQ ? 6 * N : 1];
    idx_issue(0, iw);
    idx_issue(1, iwn);
    src_issue(iw, gv); // (waits for the index words of round 0)
    coef_issue(0, cfa);
    __syncthreads();
    // one round: cf = this round's coefficients (requested a whole round ago), cfn receives the next round's
    T dummy = T(0);
    auto round = [&](int r, T(&cf)[PERQ ? 6 * N : 1], T(&cfn)[PERQ ? 6 * N : 1]) {
        T rr[N], q[N], gz[N];
        if (active) // nodal -> quadrature along x
          {
#pragma unroll
            for (int i = 0; i < N; ++i)
              rr[i] = gv[i];
          }
        // the source values of the next round (its index words were requested a round ago), index words of the round after
        src_issue(iwn, gv);
        idx_issue(r + 2, iwn);
        coef_issue(r + 1, cfn);
        if (active)
          {
            mv<N, T>(B->S, rr, q);
#pragma unroll
            for (int i = 0; i < N; ++i)
              Uc[xl + i] = q[i];
          }
        __syncthreads();
        if (active) // along y
          {
#pragma unroll
            for (int i = 0; i < N; ++i)
              rr[i] = Uc[yl + i * LN];
            mv<N, T>(B->S, rr, q);
#pragma unroll
            for (int i = 0; i < N; ++i)
              Uc[yl + i * LN] = q[i];
          }
        __syncthreads();
        if (active) // along z; z-derivative of this z-line in registers
          {
#pragma unroll
            for (int i = 0; i < N; ++i)
              rr[i] = Uc[zl + i * PL];
            mv<N, T>(B->S, rr, q);
#pragma unroll
            for (int i = 0; i < N; ++i)
              Uc[zl + i * PL] = q[i];
            mv<N, T>(B->D, q, gz);
          }
        __syncthreads();
        if (active) // x- and y-derivatives, to the z-line owners through LDS
          {
#pragma unroll
            for (int i = 0; i < N; ++i)
              q[i] = Uc[xl + i];
            mv<N, T>(B->D, q, rr);
#pragma unroll
            for (int i = 0; i < N; ++i)
              Xc[xl + i] = rr[i];
#pragma unroll
            for (int i = 0; i < N; ++i)
              q[i] = Uc[yl + i * LN];
            mv<N, T>(B->D, q, rr);
#pragma unroll
            for (int i = 0; i < N; ++i)
              Yc[yl + i * LN] = rr[i];
          }
        __syncthreads();
        if (active) // the tensor at the N points of this z-line
          {
            const T wab = B->w[a] * B->w[b];
#pragma unroll
            for (int k = 0; k < N; ++k)
              {
                T t0, t1, t2, t3, t4, t5;
                if (PERQ)
                  {
                    t0 = cf[PERQ ? 6 * k : 0], t1 = cf[PERQ ? 6 * k + 1 : 0], t2 = cf[PERQ ? 6 * k + 2 : 0];
                    t3 = cf[PERQ ? 6 * k + 3 : 0], t4 = cf[PERQ ? 6 * k + 4 : 0], t5 = cf[PERQ ? 6 * k + 5 : 0];
#ifdef MGX_GB_UNUSEDCOEF // diagnostic build (wrong results): the stream is loaded, waited for in the last phase only
                    dummy += t0 + t1 + t2 + t3 + t4 + t5;
                    t0 = t1 = t2 = T(1), t3 = t4 = t5 = T(0);
#endif
                  }
                else
                  {
                    const T w = wab * B->w[k];
                    t0 = c0 * w, t1 = c1 * w, t2 = c2 * w, t3 = c3 * w, t4 = c4 * w, t5 = c5 * w;
                  }
                const T gx = Xc[zl + k * PL], gy = Yc[zl + k * PL], g = gz[k];
                Xc[zl + k * PL] = t0 * gx + t3 * gy + t4 * g;
                Yc[zl + k * PL] = t3 * gx + t1 * gy + t5 * g;
                gz[k]           = t4 * gx + t5 * gy + t2 * g;
              }
            mvT<N, T>(B->D, gz, q);
#pragma unroll
            for (int i = 0; i < N; ++i)
              gz[i] = q[i];
          }
        __syncthreads();
        if (active) // transposed x-derivative
          {
#pragma unroll
            for (int i = 0; i < N; ++i)
              rr[i] = Xc[xl + i];
            mvT<N, T>(B->D, rr, q);
#pragma unroll
            for (int i = 0; i < N; ++i)
              Uc[xl + i] = q[i];
          }
        __syncthreads();
        if (active) // transposed y-derivative, accumulated
          {
#pragma unroll
            for (int i = 0; i < N; ++i)
              rr[i] = Yc[yl + i * LN];
            mvT<N, T>(B->D, rr, q);
#pragma unroll
            for (int i = 0; i < N; ++i)
              Uc[yl + i * LN] += q[i];
          }
        __syncthreads();
        if (active) // add the z part, quadrature -> nodal along z
          {
#pragma unroll
            for (int i = 0; i < N; ++i)
              rr[i] = Uc[zl + i * PL] + gz[i];
            mvT<N, T>(B->S, rr, q);
#pragma unroll
            for (int i = 0; i < N; ++i)
              Uc[zl + i * PL] = q[i];
          }
        __syncthreads();
        if (active) // along y
          {
#pragma unroll
            for (int i = 0; i < N; ++i)
              rr[i] = Uc[yl + i * LN];
            mvT<N, T>(B->S, rr, q);
#pragma unroll
            for (int i = 0; i < N; ++i)
              Uc[yl + i * LN] = q[i];
          }
        __syncthreads();
        if (active) // along x, added to the brick array: cells of one parity class share no point
          {
#pragma unroll
            for (int i = 0; i < N; ++i)
              rr[i] = Uc[xl + i];
            mvT<N, T>(B->S, rr, q);
            const int cx = 2 * (sl & 1) + (r & 1), cy = 2 * ((sl >> 1) & 1) + ((r >> 1) & 1), cz = 2 * (sl >> 2) + (r >> 2);
            const int pt = ((P * cz + b) * G + (P * cy + a)) * G + P * cx;
#pragma unroll
            for (int i = 0; i < N; ++i)
              W[pt + i] += q[i];
          }
        __syncthreads();
    };
#pragma unroll 1
    for (int r = 0; r < 8; r += 2)
      {
        round(r, cfa, cfb);
        round(r + 1, cfb, cfa);
      }
    if (dummy == T(12345.678))
      dst[0] = dummy;
    // write-out: items in the order of the item table (word: bits 0..9 entity slot, 10..22 brick point, 23..31 offset
    // inside the entity); constrained entities have no entry
    T *pb = priv + (size_t)brick * n_surf;
    for (int it = tid; it < NPTS; it += NT)
      {
        const uint32_t m = item_map[it], sw = m & 1023u, pnt = (m >> 10) & 8191u, off = m >> 23;
        const uint32_t w = E[sw], so = S[sw];
        if (w == kInvalid)
          continue;
        const T v = W[pnt];
        if (so != kInvalid)
          pb[so + off] = v;
        else
          dst[(w & 0x3FFFFFFFu) + off] = v;
      }
  }

  // Diagonal of the general cell matrix (local_compute_diagonal :770-800, restated in closed form):
  // d_i = sum_q C(q) : grad phi_i(q) grad phi_i(q) with grad phi_i(q) from the 1D values S and
  // nodal derivatives G = D S at the quadrature points.  One thread per cell DoF line (j, k).
  template <int P, typename T, bool PERQ>
  __global__ void __launch_bounds__(256)
    cell_diagonal_general_kernel(T *__restrict__ diag, const uint32_t *__restrict__ idx27, uint32_t n_cells,
                                 const Basis1D<T> *__restrict__ B, const T *__restrict__ G1,
                                 const T *__restrict__ coef_q, T c0, T c1, T c2, T c3, T c4, T c5,
                                 const uint32_t *__restrict__ cell_list, T *__restrict__ scratch)
  {
    // cell_list: the launch covers cells of one colour (no shared DoF), plain adds; scratch: local
    // results for the ordered assembly; neither: atomic adds (non-reproducible last bits, fallback)
    constexpr int  N = P + 1, N3 = N * N * N;
    const uint32_t gid  = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t pos  = gid / (N * N);
    if (pos >= n_cells)
      return;
    const uint32_t cell = cell_list ? cell_list[pos] : pos;
    const int t = gid % (N * N), j = t % N, k = t / N;
    T         r[N];
#pragma unroll
    for (int i = 0; i < N; ++i)
      r[i] = T(0);
    const T *cq = coef_q + (size_t)cell * 6 * N3;
    for (int qz = 0; qz < N; ++qz)
      for (int qy = 0; qy < N; ++qy)
        {
          const T sy = B->S[qy * N + j], sz = B->S[qz * N + k], gy1 = G1[qy * N + j], gz1 = G1[qz * N + k];
          for (int qx = 0; qx < N; ++qx)
            {
              const int q = (qz * N + qy) * N + qx;
              T         t0, t1, t2, t3, t4, t5;
              if (PERQ)
                t0 = cq[q], t1 = cq[N3 + q], t2 = cq[2 * N3 + q], t3 = cq[3 * N3 + q], t4 = cq[4 * N3 + q], t5 = cq[5 * N3 + q];
              else
                {
                  const T w = B->w[qx] * B->w[qy] * B->w[qz];
                  t0 = c0 * w, t1 = c1 * w, t2 = c2 * w, t3 = c3 * w, t4 = c4 * w, t5 = c5 * w;
                }
#pragma unroll
              for (int i = 0; i < N; ++i)
                {
                  const T dx = G1[qx * N + i] * sy * sz, dy = B->S[qx * N + i] * gy1 * sz, dz = B->S[qx * N + i] * sy * gz1;
                  r[i] += t0 * dx * dx + t1 * dy * dy + t2 * dz * dz + T(2) * (t3 * dx * dy + t4 * dx * dz + t5 * dy * dz);
                }
            }
        }
    if (scratch)
      return store_line_local<P, T>(scratch, cell, j, k, r);
    const LineIndex<P> L = line_index<P>(idx27, cell, j, k);
    if (cell_list)
      scatter_add_line_plain<P, T>(diag, L, r);
    else
      scatter_add_line<P, T>(diag, L, r);
  }

  // diagonal of the cell matrix (local_compute_diagonal, laplace_operator.h:770-800).  For the
  // affine constant-coefficient tensor the unit-vector applications collapse to
  //   d_i = c0 a[ix] m[iy] m[iz] + c1 m[ix] a[iy] m[iz] + c2 m[ix] m[iy] a[iz]
  // with a[i] = sum_q w_q (dphi_i(x_q))^2, m[i] = sum_q w_q phi_i(x_q)^2.
  template <typename T>
  struct Diag1D
  {
    T a[kMaxN], m[kMaxN];
  };

  template <int P, typename T>
  __global__ void __launch_bounds__(256)
    cell_diagonal_kernel(T *__restrict__ diag, const uint32_t *__restrict__ idx27, uint32_t n_cells,
                         Diag1D<T> d1, T c0, T c1, T c2, const uint32_t *__restrict__ cell_list, T *__restrict__ scratch)
  {
    constexpr int  N    = P + 1;
    const uint32_t gid  = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t pos  = gid / (N * N);
    if (pos >= n_cells)
      return;
    const uint32_t cell = cell_list ? cell_list[pos] : pos;
    const int t = gid % (N * N), j = t % N, k = t / N;
    T         r[N];
#pragma unroll
    for (int i = 0; i < N; ++i)
      r[i] = c0 * d1.a[i] * d1.m[j] * d1.m[k] + c1 * d1.m[i] * d1.a[j] * d1.m[k] +
             c2 * d1.m[i] * d1.m[j] * d1.a[k];
    if (scratch)
      return store_line_local<P, T>(scratch, cell, j, k, r);
    const LineIndex<P> L = line_index<P>(idx27, cell, j, k);
    if (cell_list)
      scatter_add_line_plain<P, T>(diag, L, r);
    else
      scatter_add_line<P, T>(diag, L, r);
  }

  // ------------------------------------------------------------------------------------------
  // Transfers (MGTransferMatrixFree restated): one workgroup per parent
  // cell; the (p+1)^3 coarse values are interpolated to the (2p+1)^3 points of the children
  // patch by three LDS-staged sweeps with P1.
  // ------------------------------------------------------------------------------------------
  template <int P>
  struct TCfg
  {
    static constexpr int N       = P + 1;
    static constexpr int M       = 2 * P + 1;
    static constexpr int THREADS = ((M * M + 63) / 64) * 64 > 1024 ? 1024 : ((M * M + 63) / 64) * 64;
  };

  template <int P>
  __device__ __forceinline__ int patch_code(int a)
  {
    return a == 0 ? 0 : (a == 2 * P ? 2 : 1);
  }

  template <int P, typename T>
  __global__ void __launch_bounds__(TCfg<P>::THREADS)
    prolongate_kernel(T *__restrict__ fine, const T *__restrict__ coarse, const uint32_t *__restrict__ idx_c,
                      const uint32_t *__restrict__ idx_f, const uint32_t *__restrict__ children,
                      const uint32_t *__restrict__ own27, uint32_t n_parents, const Basis1D<T> *__restrict__ B,
                      int add)
  {
    constexpr int N = P + 1, M = 2 * P + 1;
    __shared__ T  p1[M * N];
    __shared__ T  in[N * N * N];
    __shared__ T  t1[N * N * M];
    __shared__ T  t2[N * M * M];
    __shared__ T  out[M * M * M];
    const uint32_t pc = blockIdx.x;
    if (pc >= n_parents)
      return;
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int i = tid; i < M * N; i += nt)
      p1[i] = B->P1[i];
    for (int t = tid; t < N * N; t += nt)
      {
        const int    j = t % N, k = t / N;
        T            r[N];
        LineIndex<P> L = line_index<P>(idx_c, pc, j, k);
        gather_line<P, T>(coarse, L, r);
#pragma unroll
        for (int i = 0; i < N; ++i)
          in[(k * N + j) * N + i] = r[i];
      }
    __syncthreads();
    for (int o = tid; o < N * N * M; o += nt) // x: [k][j][a]
      {
        const int a = o % M, kj = o / M;
        T         s = 0;
#pragma unroll
        for (int i = 0; i < N; ++i)
          s = fma(p1[a * N + i], in[kj * N + i], s);
        t1[o] = s;
      }
    __syncthreads();
    for (int o = tid; o < N * M * M; o += nt) // y: [k][b][a]
      {
        const int a = o % M, b = (o / M) % M, k = o / (M * M);
        T         s = 0;
#pragma unroll
        for (int j = 0; j < N; ++j)
          s = fma(p1[b * N + j], t1[(k * N + j) * M + a], s);
        t2[o] = s;
      }
    __syncthreads();
    for (int o = tid; o < M * M * M; o += nt) // z: [c][b][a]
      {
        const int ba = o % (M * M), c = o / (M * M);
        T         s = 0;
#pragma unroll
        for (int k = 0; k < N; ++k)
          s = fma(p1[c * N + k], t2[k * M * M + ba], s);
        out[o] = s;
      }
    __syncthreads();
    // distribute to the 8 children.  Every fine DoF is written by exactly one fine cell, the first
    // cell (in cell order) that contains its entity (own27 bit mask, computed on the host from
    // the index table).  The values the different parents/children compute for a shared DoF are
    // bitwise identical (the 1D prolongation rows at coinciding nodes are exact unit vectors),
    // so "owner writes once" equals deal.II's "every cell adds value/multiplicity" -- without
    // atomics and bitwise reproducibly.
    for (int w = tid; w < 8 * N * N; w += nt)
      {
        const int      ch = w / (N * N), t = w % (N * N), j = t % N, k = t / N;
        const uint32_t fc = children[8u * (size_t)pc + ch];
        const int      ox = (ch & 1) * P, oy = ((ch >> 1) & 1) * P, oz = (ch >> 2) * P;
        const int      b = oy + j, c = oz + k;
        LineIndex<P>   L = line_index<P>(idx_f, fc, j, k);
        int            cy, o1, cz, o2;
        node_code<P>(j, cy, o1);
        node_code<P>(k, cz, o2);
        const uint32_t own = own27[fc] >> (9 * cz + 3 * cy); // bits 0,1,2: left, interior, right
        T              r[N];
#pragma unroll
        for (int i = 0; i < N; ++i)
          r[i] = out[(c * M + b) * M + ox + i];
        if (L.b0 != kInvalid && (own & 1u))
          {
            T *p = fine + L.b0 + L.off;
            *p   = add ? *p + r[0] : r[0];
          }
        if (L.b1 != kInvalid && (own & 2u))
          {
#pragma unroll
            for (int i = 0; i < P - 1; ++i)
              {
                T *p = fine + L.b1 + L.off * (uint32_t)(P - 1) + (uint32_t)i;
                *p   = add ? *p + r[1 + i] : r[1 + i];
              }
          }
        if (L.b2 != kInvalid && (own & 4u))
          {
            T *p = fine + L.b2 + L.off;
            *p   = add ? *p + r[P] : r[P];
          }
      }
  }

  template <int P, typename T>
  __global__ void __launch_bounds__(TCfg<P>::THREADS)
    restrict_kernel(T *__restrict__ coarse, const T *__restrict__ fine, const uint32_t *__restrict__ idx_c,
                    const uint32_t *__restrict__ idx_f, const uint32_t *__restrict__ children,
                    const uint8_t *__restrict__ wshift, uint32_t n_parents, const Basis1D<T> *__restrict__ B)
  {
    constexpr int N = P + 1, M = 2 * P + 1;
    __shared__ T  p1[M * N];
    __shared__ T  in[N * N * N];
    __shared__ T  t1[N * N * M];
    __shared__ T  t2[N * M * M];
    __shared__ T  out[M * M * M];
    const uint32_t pc = blockIdx.x;
    if (pc >= n_parents)
      return;
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int i = tid; i < M * N; i += nt)
      p1[i] = B->P1[i];
    // gather the weighted fine patch (shared points are written with identical values by the
    // children that hold them)
    for (int w = tid; w < 8 * N * N; w += nt)
      {
        const int      ch = w / (N * N), t = w % (N * N), j = t % N, k = t / N;
        const uint32_t fc = children[8u * (size_t)pc + ch];
        const int      ox = (ch & 1) * P, oy = ((ch >> 1) & 1) * P, oz = (ch >> 2) * P;
        const int      b = oy + j, c = oz + k;
        LineIndex<P>   L = line_index<P>(idx_f, fc, j, k);
        T              r[N];
        gather_line<P, T>(fine, L, r);
        const uint8_t *ws = wshift + 27u * (size_t)pc + 9 * patch_code<P>(c) + 3 * patch_code<P>(b);
#pragma unroll
        for (int i = 0; i < N; ++i)
          {
            const int sh               = ws[patch_code<P>(ox + i)];
            out[(c * M + b) * M + ox + i] = r[i] * (T(1) / T(1 << sh));
          }
      }
    __syncthreads();
    for (int o = tid; o < N * M * M; o += nt) // z^T: [k][b][a]
      {
        const int ba = o % (M * M), k = o / (M * M);
        T         s = 0;
#pragma unroll
        for (int c = 0; c < M; ++c)
          s = fma(p1[c * N + k], out[c * M * M + ba], s);
        t2[o] = s;
      }
    __syncthreads();
    for (int o = tid; o < N * N * M; o += nt) // y^T: [k][j][a]
      {
        const int a = o % M, j = (o / M) % N, k = o / (M * N);
        T         s = 0;
#pragma unroll
        for (int b = 0; b < M; ++b)
          s = fma(p1[b * N + j], t2[(k * M + b) * M + a], s);
        t1[o] = s;
      }
    __syncthreads();
    for (int o = tid; o < N * N * N; o += nt) // x^T: [k][j][i]
      {
        const int i = o % N, kj = o / N;
        T         s = 0;
#pragma unroll
        for (int a = 0; a < M; ++a)
          s = fma(p1[a * N + i], t1[kj * M + a], s);
        in[o] = s;
      }
    __syncthreads();
    for (int t = tid; t < N * N; t += nt)
      {
        const int    j = t % N, k = t / N;
        LineIndex<P> L = line_index<P>(idx_c, pc, j, k);
        T            r[N];
#pragma unroll
        for (int i = 0; i < N; ++i)
          r[i] = in[(k * N + j) * N + i];
        scatter_add_line<P, T>(coarse, L, r);
      }
  }

  // ------------------------------------------------------------------------------------------
  // launchers
  // ------------------------------------------------------------------------------------------
#define MGX_DISPATCH_P(p, ...)                 \
  switch (p)                                   \
    {                                          \
      case 1: { constexpr int P = 1; __VA_ARGS__; } break; \
      case 2: { constexpr int P = 2; __VA_ARGS__; } break; \
      case 3: { constexpr int P = 3; __VA_ARGS__; } break; \
      case 4: { constexpr int P = 4; __VA_ARGS__; } break; \
      case 5: { constexpr int P = 5; __VA_ARGS__; } break; \
      case 6: { constexpr int P = 6; __VA_ARGS__; } break; \
      case 7: { constexpr int P = 7; __VA_ARGS__; } break; \
      case 8: { constexpr int P = 8; __VA_ARGS__; } break; \
      case 9: { constexpr int P = 9; __VA_ARGS__; } break; \
      default: break;                          \
    }

  template <typename T>
  static void assemble_t(hipStream_t s, const OperatorData &op, int mode, void *dst, const void *tail_src, uint32_t n_head)
  {
    const uint32_t nb = (op.n_dofs + 255) / 256;
    if (mode == 1)
      hipLaunchKernelGGL((assemble_kernel<T, 1>), dim3(nb), dim3(256), 0, s, (T *)dst, (const T *)op.cell_scratch,
                         op.asm_start, op.asm_pos, op.n_dofs, (const T *)nullptr, 0u);
    else
      hipLaunchKernelGGL((assemble_kernel<T, 0>), dim3(nb), dim3(256), 0, s, (T *)dst, (const T *)op.cell_scratch,
                         op.asm_start, op.asm_pos, op.n_dofs, (const T *)tail_src, n_head);
  }

  void launch_assemble(hipStream_t s, const OperatorData &op, int mode, void *dst, const void *tail_src, uint32_t n_head)
  {
    if (op.number == 1)
      assemble_t<double>(s, op, mode, dst, tail_src, n_head);
    else
      assemble_t<float>(s, op, mode, dst, tail_src, n_head);
  }

  // Three ways to add the cell contributions up (op decides): ordered assembly through the scratch
  // array (op.asm_start; dst is written, not added to), one launch per cell colour with plain
  // read-modify-writes (op.cell_order), or one launch with atomic adds (fallback).
  template <int P, typename T>
  static void cell_loop_t(hipStream_t s, const OperatorData &op, void *dst, const void *src, const void *tail_src,
                          uint32_t n_head, const ChebPost *post)
  {
    using C            = Cfg<P>;
    const uint32_t nb  = (op.n_cells + C::CPB - 1) / C::CPB;
    T             *scratch = op.asm_start ? (T *)op.cell_scratch : nullptr;
    if (op.coef_q || op.full_tensor)
      {
        using C = Cfg<P, MGX_GENERAL_WG_THREADS>;
        const bool coloured = op.cell_order && !scratch;
        const int  n_launch = coloured ? op.n_cell_colours : 1;
        for (int k = 0; k < n_launch; ++k)
          {
            const uint32_t  first = coloured ? op.cell_colour_start[k] : 0;
            const uint32_t  count = coloured ? op.cell_colour_start[k + 1] - first : op.n_cells;
            const uint32_t *list  = coloured ? op.cell_order + first : nullptr;
            const uint32_t  nbk   = (count + C::CPB - 1) / C::CPB;
            if (count == 0)
              continue;
            if (op.coef_q)
              hipLaunchKernelGGL((cell_loop_general_kernel<P, T, true>), dim3(nbk), dim3(C::THREADS), 0, s, (T *)dst,
                                 (const T *)src, op.idx27, count, (const Basis1D<T> *)op.basis, (const T *)op.coef_q, (T)0,
                                 (T)0, (T)0, (T)0, (T)0, (T)0, list, scratch);
            else
              hipLaunchKernelGGL((cell_loop_general_kernel<P, T, false>), dim3(nbk), dim3(C::THREADS), 0, s, (T *)dst,
                                 (const T *)src, op.idx27, count, (const Basis1D<T> *)op.basis, (const T *)nullptr,
                                 (T)op.coef[0], (T)op.coef[1], (T)op.coef[2], (T)op.coef[3], (T)op.coef[4], (T)op.coef[5],
                                 list, scratch);
          }
      }
    else
      hipLaunchKernelGGL((cell_loop_kernel<P, T>), dim3(nb), dim3(C::THREADS), 0, s, (T *)dst, (const T *)src,
                         op.idx27, op.n_cells, (const Basis1D<T> *)op.basis, (T)op.coef[0], (T)op.coef[1],
                         (T)op.coef[2], scratch);
    if (scratch && post)
      {
        const uint32_t nba = (op.n_dofs + 255) / 256;
        if (post->three_term)
          hipLaunchKernelGGL((assemble_cheb_kernel<T, true>), dim3(nba), dim3(256), 0, s, (T *)dst, (const T *)scratch, op.asm_start,
                             op.asm_pos, op.n_dofs, n_head, (T *)post->x_old, (const T *)post->b, (const T *)post->dinv,
                             (T)post->f1, (T)post->f2);
        else
          hipLaunchKernelGGL((assemble_cheb_kernel<T, false>), dim3(nba), dim3(256), 0, s, (T *)dst, (const T *)scratch, op.asm_start,
                             op.asm_pos, op.n_dofs, n_head, (T *)post->x_old, (const T *)post->b, (const T *)post->dinv,
                             (T)post->f1, (T)post->f2);
      }
    else if (scratch)
      assemble_t<T>(s, op, 0, dst, tail_src, n_head);
  }

  // ------------------------------------------------------------------------------------------
  // Transfer between the DG space and the FE_Q space of the same mesh and degree, cell by cell
  // (laplace_operator_dg.h:1798-1819 residual -> FE_Q, :1863-1894 FE_Q -> DG).  Both spaces contain
  // Q_p on every cell: the embedding of an FE_Q function is exact, d = (P1 x P1 x P1) c with the 1D
  // matrix P1 = (phi_i(g_q))^-1 from the values in the Gauss-Lobatto nodes g_q to the coefficients of
  // the DG basis phi_i; the restriction is its transpose.  The DG vector holds (p+1)^3 contiguous
  // values per cell, cells in the order of the compressed index table.
  // TO_DG: dg[cell] += P c[cell]   else: cg += P^T dg[cell] (atomics; constrained rows skipped)
  // ------------------------------------------------------------------------------------------
  template <int P, typename T, bool TO_DG>
  __global__ void __launch_bounds__(Cfg<P>::THREADS)
    dg_cg_transfer_kernel(T *__restrict__ dst, const T *__restrict__ src, const uint32_t *__restrict__ idx27,
                          uint32_t n_cells, const T *__restrict__ P1, uint32_t cell_first, uint32_t cell_stride, int plain)
  {
    // cells cell_first + cell_stride * i, i < n_cells; plain: no two cells of the launch share a DoF,
    // the restriction adds without atomics
    using C         = Cfg<P>;
    constexpr int N = C::N, LN = C::LN, N3 = N * N * N;
    __shared__ T  tile[C::CPB * C::CELL_LDS];
    const int     tid = threadIdx.x, cw = tid / C::TPC, t = tid - cw * C::TPC;
    const int     a = t % N, b = t / N;
    const bool    lane_ok = cw < C::CPB;
    uint32_t      cell    = blockIdx.x * C::CPB + (lane_ok ? cw : 0);
    const bool    active  = lane_ok && cell < n_cells;
    if (cell >= n_cells)
      cell = n_cells - 1;
    cell = cell_first + cell_stride * cell;
    T *U = tile + (lane_ok ? cw : 0) * C::CELL_LDS;
    T  r[N], o[N];
    // x-lines (j = a, k = b)
    if (lane_ok)
      {
        if (TO_DG)
          {
            const LineIndex<P> L = line_index<P>(idx27, cell, a, b);
            gather_line<P, T>(src, L, r);
            mv<N>(P1, r, o);
          }
        else
          {
#pragma unroll
            for (int i = 0; i < N; ++i)
              r[i] = src[(size_t)cell * N3 + (b * N + a) * N + i];
            mvT<N>(P1, r, o);
          }
#pragma unroll
        for (int i = 0; i < N; ++i)
          U[(b * N + a) * LN + i] = o[i];
      }
    __syncthreads();
    if (lane_ok) // y-lines (i = a, k = b)
      {
#pragma unroll
        for (int j = 0; j < N; ++j)
          r[j] = U[(b * N + j) * LN + a];
        if (TO_DG)
          mv<N>(P1, r, o);
        else
          mvT<N>(P1, r, o);
#pragma unroll
        for (int j = 0; j < N; ++j)
          U[(b * N + j) * LN + a] = o[j];
      }
    __syncthreads();
    if (lane_ok) // z-lines (i = a, j = b)
      {
#pragma unroll
        for (int k = 0; k < N; ++k)
          r[k] = U[(k * N + b) * LN + a];
        if (TO_DG)
          mv<N>(P1, r, o);
        else
          mvT<N>(P1, r, o);
#pragma unroll
        for (int k = 0; k < N; ++k)
          U[(k * N + b) * LN + a] = o[k];
      }
    __syncthreads();
    if (active)
      {
#pragma unroll
        for (int i = 0; i < N; ++i)
          r[i] = U[(b * N + a) * LN + i];
        if (TO_DG)
          {
#pragma unroll
            for (int i = 0; i < N; ++i)
              dst[(size_t)cell * N3 + (b * N + a) * N + i] += r[i];
          }
        else if (plain)
          scatter_add_line_plain<P, T>(dst, line_index<P>(idx27, cell, a, b), r);
        else
          scatter_add_line<P, T>(dst, line_index<P>(idx27, cell, a, b), r);
      }
  }

  template <int P, typename T>
  static void dg_cg_transfer_t(hipStream_t s, bool to_dg, void *dst, const void *src, const uint32_t *idx27,
                               uint32_t n_cells, const void *P1, bool eight_colours)
  {
    using C = Cfg<P>;
    if (to_dg)
      {
        const uint32_t nb = (n_cells + C::CPB - 1) / C::CPB;
        hipLaunchKernelGGL((dg_cg_transfer_kernel<P, T, true>), dim3(nb), dim3(C::THREADS), 0, s, (T *)dst, (const T *)src,
                           idx27, n_cells, (const T *)P1, 0u, 1u, 0);
        return;
      }
    // restriction: cells c, c + 8, c + 16, ... share no DoF when the caller says so (forest order: the same
    // child of every parent) -- eight launches with plain read-modify-writes instead of one with atomics
    const uint32_t stride = eight_colours ? 8u : 1u;
    for (uint32_t k = 0; k < stride; ++k)
      {
        const uint32_t count = (n_cells - k + stride - 1) / stride;
        if (k >= n_cells || count == 0)
          continue;
        const uint32_t nb = (count + C::CPB - 1) / C::CPB;
        hipLaunchKernelGGL((dg_cg_transfer_kernel<P, T, false>), dim3(nb), dim3(C::THREADS), 0, s, (T *)dst, (const T *)src,
                           idx27, count, (const T *)P1, k, stride, eight_colours ? 1 : 0);
      }
  }

  void launch_dg_cg_transfer(hipStream_t s, int number, int p, bool to_dg, void *dst, const void *src,
                             const uint32_t *idx27, uint32_t n_cells, const void *P1, bool eight_colours)
  {
    if (n_cells == 0)
      return;
    if (number == 1)
      {
        MGX_DISPATCH_P(p, dg_cg_transfer_t<P, double>(s, to_dg, dst, src, idx27, n_cells, P1, eight_colours));
      }
    else
      {
        MGX_DISPATCH_P(p, dg_cg_transfer_t<P, float>(s, to_dg, dst, src, idx27, n_cells, P1, eight_colours));
      }
  }

  void launch_cell_loop(hipStream_t s, const OperatorData &op, void *dst, const void *src, const void *tail_src,
                        uint32_t n_head, const ChebPost *post)
  {
    if (op.number == 1)
      {
        MGX_DISPATCH_P(op.p, cell_loop_t<P, double>(s, op, dst, src, tail_src, n_head, post));
      }
    else
      {
        MGX_DISPATCH_P(op.p, cell_loop_t<P, float>(s, op, dst, src, tail_src, n_head, post));
      }
  }

  // the general operator on its brick schedule (brick_general_kernel): interior DoFs to dst, the bricks' private
  // blocks to op.gbricks.fr.priv; the caller completes the surface DoFs with launch_surf_finish on that schedule
  void launch_general_bricks(hipStream_t s, const OperatorData &op, void *dst, const void *src)
  {
    const BrickData &g = op.gbricks;
    auto run = [&](auto number) {
      using T = decltype(number);
      if (op.coef_q)
        hipLaunchKernelGGL((brick_general_kernel<4, T, true>), dim3(g.n_bricks), dim3(256), 0, s, (T *)dst, (const T *)src, op.idx27,
                           g.fr.ent, g.fr.surf_off, g.item_map, g.order_dev, g.n_bricks, (const Basis1D<T> *)op.basis, (const T *)op.coef_q,
                           (T)0, (T)0, (T)0, (T)0, (T)0, (T)0, (T *)g.fr.priv, g.fr.n_surf, (uint32_t)((size_t)op.n_dofs * sizeof(T)));
      else
        hipLaunchKernelGGL((brick_general_kernel<4, T, false>), dim3(g.n_bricks), dim3(256), 0, s, (T *)dst, (const T *)src, op.idx27,
                           g.fr.ent, g.fr.surf_off, g.item_map, g.order_dev, g.n_bricks, (const Basis1D<T> *)op.basis, (const T *)nullptr,
                           (T)op.coef[0], (T)op.coef[1], (T)op.coef[2], (T)op.coef[3], (T)op.coef[4], (T)op.coef[5],
                           (T *)g.fr.priv, g.fr.n_surf, (uint32_t)((size_t)op.n_dofs * sizeof(T)));
    };
    if (op.number == 1)
      run(double());
    else
      run(float());
  }

  // lists[k] .. lists[k + 1]: device cell lists of n_lists launches whose cells share no DoF (plain
  // adds); n_lists == 0: ordered assembly if the operator has the tables, else one launch with atomics
  template <int P, typename T>
  static void cell_diag_t(hipStream_t s, const OperatorData &op, void *diag, const double *a, const double *m,
                          const uint32_t *lists, const uint32_t *list_start, int n_lists)
  {
    constexpr int N = P + 1;
    Diag1D<T>     d1;
    for (int i = 0; i < N; ++i)
      {
        d1.a[i] = (T)a[i];
        d1.m[i] = (T)m[i];
      }
    T        *scratch  = (n_lists == 0 && op.asm_start) ? (T *)op.cell_scratch : nullptr;
    const int n_launch = n_lists > 0 ? n_lists : 1;
    for (int k = 0; k < n_launch; ++k)
      {
        const uint32_t  count = n_lists > 0 ? list_start[k + 1] - list_start[k] : op.n_cells;
        const uint32_t *list  = n_lists > 0 ? lists + list_start[k] : nullptr;
        if (count == 0)
          continue;
        const uint64_t nthreads = (uint64_t)count * N * N;
        const uint32_t nb       = (uint32_t)((nthreads + 255) / 256);
        if (op.coef_q)
          hipLaunchKernelGGL((cell_diagonal_general_kernel<P, T, true>), dim3(nb), dim3(256), 0, s, (T *)diag, op.idx27, count,
                             (const Basis1D<T> *)op.basis, (const T *)op.grad_1d, (const T *)op.coef_q, (T)0, (T)0, (T)0,
                             (T)0, (T)0, (T)0, list, scratch);
        else if (op.full_tensor)
          hipLaunchKernelGGL((cell_diagonal_general_kernel<P, T, false>), dim3(nb), dim3(256), 0, s, (T *)diag, op.idx27, count,
                             (const Basis1D<T> *)op.basis, (const T *)op.grad_1d, (const T *)nullptr, (T)op.coef[0],
                             (T)op.coef[1], (T)op.coef[2], (T)op.coef[3], (T)op.coef[4], (T)op.coef[5], list, scratch);
        else
          hipLaunchKernelGGL((cell_diagonal_kernel<P, T>), dim3(nb), dim3(256), 0, s, (T *)diag, op.idx27, count, d1,
                             (T)op.coef[0], (T)op.coef[1], (T)op.coef[2], list, scratch);
      }
    if (scratch)
      assemble_t<T>(s, op, 0, diag, nullptr, 0u);
  }

  void launch_cell_diagonal(hipStream_t s, const OperatorData &op, void *diag, const double *a, const double *m,
                            const uint32_t *lists, const uint32_t *list_start, int n_lists)
  {
    if (op.number == 1)
      {
        MGX_DISPATCH_P(op.p, cell_diag_t<P, double>(s, op, diag, a, m, lists, list_start, n_lists));
      }
    else
      {
        MGX_DISPATCH_P(op.p, cell_diag_t<P, float>(s, op, diag, a, m, lists, list_start, n_lists));
      }
  }

  // compute_residual (cell_loop_general_kernel, RESID) with the assembly variants of the diagonal above
  template <int P, typename T>
  static void cell_residual_t(hipStream_t s, const OperatorData &op, void *dst, const void *src, const void *rhs_q,
                              const uint32_t *lists, const uint32_t *list_start, int n_lists)
  {
    using C            = Cfg<P, MGX_GENERAL_WG_THREADS>;
    T        *scratch  = (n_lists == 0 && op.asm_start) ? (T *)op.cell_scratch : nullptr;
    const int n_launch = n_lists > 0 ? n_lists : 1;
    for (int k = 0; k < n_launch; ++k)
      {
        const uint32_t  count = n_lists > 0 ? list_start[k + 1] - list_start[k] : op.n_cells;
        const uint32_t *list  = n_lists > 0 ? lists + list_start[k] : nullptr;
        if (count == 0)
          continue;
        const uint32_t nb = (count + C::CPB - 1) / C::CPB;
        if (op.coef_q)
          hipLaunchKernelGGL((cell_loop_general_kernel<P, T, true, true>), dim3(nb), dim3(C::THREADS), 0, s, (T *)dst, (const T *)src,
                             op.idx27, count, (const Basis1D<T> *)op.basis, (const T *)op.coef_q, (T)0, (T)0, (T)0, (T)0, (T)0, (T)0,
                             list, scratch, op.idx27_plain, (const T *)rhs_q);
        else
          hipLaunchKernelGGL((cell_loop_general_kernel<P, T, false, true>), dim3(nb), dim3(C::THREADS), 0, s, (T *)dst,
                             (const T *)src, op.idx27, count, (const Basis1D<T> *)op.basis, (const T *)nullptr, (T)op.coef[0],
                             (T)op.coef[1], (T)op.coef[2], (T)op.coef[3], (T)op.coef[4], (T)op.coef[5], list, scratch,
                             op.idx27_plain, (const T *)rhs_q);
      }
    if (scratch)
      assemble_t<T>(s, op, 0, dst, nullptr, 0u);
  }

  void launch_cell_residual(hipStream_t s, const OperatorData &op, void *dst, const void *src, const void *rhs_q,
                            const uint32_t *lists, const uint32_t *list_start, int n_lists)
  {
    if (op.number == 1)
      {
        MGX_DISPATCH_P(op.p, cell_residual_t<P, double>(s, op, dst, src, rhs_q, lists, list_start, n_lists));
      }
    else
      {
        MGX_DISPATCH_P(op.p, cell_residual_t<P, float>(s, op, dst, src, rhs_q, lists, list_start, n_lists));
      }
  }

  template <int P, typename T>
  static void prolongate_t(hipStream_t s, const TransferData &t, void *fine, const void *coarse, bool add,
                           bool with_constraints)
  {
    const OperatorData &c = *t.coarse, &f = *t.fine;
    const uint32_t     *idx_c = with_constraints ? c.idx27 : c.idx27_plain;
    hipLaunchKernelGGL((prolongate_kernel<P, T>), dim3(c.n_cells), dim3(TCfg<P>::THREADS), 0, s, (T *)fine,
                       (const T *)coarse, idx_c, f.idx27_plain, t.children, t.own27, c.n_cells,
                       (const Basis1D<T> *)c.basis, add ? 1 : 0);
  }

  void launch_prolongate(hipStream_t s, const TransferData &t, void *fine, const void *coarse, bool add,
                         bool with_constraints)
  {
    if (t.patch)
      return launch_prolongate_pipe(s, t, fine, coarse, add, with_constraints);
    if (t.coarse->number == 1)
      {
        MGX_DISPATCH_P(t.coarse->p, prolongate_t<P, double>(s, t, fine, coarse, add, with_constraints));
      }
    else
      {
        MGX_DISPATCH_P(t.coarse->p, prolongate_t<P, float>(s, t, fine, coarse, add, with_constraints));
      }
  }

  template <int P, typename T>
  static void restrict_t(hipStream_t s, const TransferData &t, void *coarse, const void *fine,
                         bool with_constraints)
  {
    const OperatorData &c = *t.coarse, &f = *t.fine;
    const uint32_t     *idx_c = with_constraints ? c.idx27 : c.idx27_plain;
    hipLaunchKernelGGL((restrict_kernel<P, T>), dim3(c.n_cells), dim3(TCfg<P>::THREADS), 0, s, (T *)coarse,
                       (const T *)fine, idx_c, f.idx27_plain, t.children, t.weight_shift, c.n_cells,
                       (const Basis1D<T> *)c.basis);
  }

  void launch_restrict_add(hipStream_t s, const TransferData &t, void *coarse, const void *fine,
                           bool with_constraints)
  {
    if (t.patch)
      return launch_restrict_add_pipe(s, t, coarse, fine, with_constraints);
    if (t.coarse->number == 1)
      {
        MGX_DISPATCH_P(t.coarse->p, restrict_t<P, double>(s, t, coarse, fine, with_constraints));
      }
    else
      {
        MGX_DISPATCH_P(t.coarse->p, restrict_t<P, float>(s, t, coarse, fine, with_constraints));
      }
  }
} // namespace mgx
