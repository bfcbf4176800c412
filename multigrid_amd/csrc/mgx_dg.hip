// mgx_dg.hip -- DG (symmetric interior penalty) Laplace operator on an affine mesh with the merged
// Chebyshev update: C ABI of include/mgx_dg.h, host-side 1D setup and the gfx950 cell kernel.
//
// Reference behaviour (not code): common/laplace_operator_dg.h -- LaplaceOperatorCompactCombine
// :350-2024 (cell-based loop operation_on_cells :1110-1861), JacobiTransformed :2028-2256,
// LocalBasisTransformer :92-350; 1D line kernel with face values common/matrix_vector_kernel.h
// :30-216.  The bilinear form is the one of common/laplace_operator_dg_face.h:66-160.
//
// Design (MI355X): a workgroup of 128 threads takes CPW = 128 / (p+1)^2 consecutive cells; the
// (p+1)^2 threads of a cell each own one line of the cell per sweep direction (registers) and one
// quadrature point of each of the 6 faces.  Per cell the LDS holds the values U in the Gauss
// points, two gradient components, and four (p+1)^2 arrays per face (own trace, own normal
// derivative, neighbour trace, neighbour normal derivative), which the face phase turns in place
// into the two arrays the integration needs.  The traces of a line's two end faces fall out of
// the sweep that has the line in registers (the `do_dg` idea of matrix_vector_kernel.h:114-141).
// Neighbour data is read straight from the source vector: two node layers per face for the
// Hermite-like basis (laplace_operator_dg.h:1359-1457), a contracted full cell otherwise.  The
// inverse diagonal of the block-Jacobi preconditioner depends only on which faces of a cell are
// Dirichlet faces: a table of at most 64 x (p+1)^3 values replaces the reference's per-cell
// stream, so the fused Chebyshev step moves 4 vector accesses per DoF (source, right-hand side,
// old iterate, new iterate) where the reference's model counts 5 (matvec_dg_cheby/program.cc:178).
#include "mgx_internal.hpp"

#include "../../include/mgx_dg.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <numeric>
#include <string>
#include <vector>

namespace
{
  using mgx::kMaxN;

  // ------------------------------------------------------------------------------------------
  // device data
  // Even-odd form of an n x n matrix M with M[q][i] = s M[n-1-q][n-1-i] (s = +1: values of a
  // symmetric basis in symmetric points; s = -1: derivatives), H = n / 2:
  //   eo[q H + i] = { (M[q][i] + M[n-1-q][i]) / 2, (M[q][i] - M[n-1-q][i]) / 2 }   (q, i < H)
  //   mrow[i] = M[H][i], mcol[q] = M[q][H], mm = M[H][H]                           (n odd)
  template <typename T>
  struct EOLine
  {
    T eo[2 * (kMaxN / 2) * (kMaxN / 2)];
    T mrow[kMaxN / 2], mcol[kMaxN / 2], mm;
  };

  template <typename T>
  struct DGConst
  {
    T S[kMaxN * kMaxN];  // S[q*n+i]   element basis -> values in the Gauss points
    T D[kMaxN * kMaxN];  // D[q*n+r]   derivative of the Gauss-point Lagrange basis in the Gauss points
    T E[kMaxN * kMaxN];  // E[i*n+e]   eigenvector e of (Laplace + penalty, mass) in the element basis
    T St[kMaxN * kMaxN], Dt[kMaxN * kMaxN], Et[kMaxN * kMaxN]; // their transposes (mul() reads these)
    // even-odd form of S, St (symmetric under reversal of both indices) and D, Dt (antisymmetric),
    // matrix_vector_kernel.h:47-113; see mul_eo below
    EOLine<T> eoS, eoSt, eoD, eoDt;
    // eigenvectors sorted by parity (even ones first): E[n-1-i][e] = +-E[i][e].  With H = n / 2, Ne = n - H
    // even and No = H odd ones: epair[i No + k] = {E[i][k], E[i][Ne + k]} (i < H, k < No),
    // elast[i] = E[i][Ne - 1] and emid[e] = E[H][e] (n odd).  (eo_e: the host found the parities pure; it refuses
    // the operator otherwise -- a run-time choice between the two forms in the kernel costs 10-15 %)
    T   epair[2 * (kMaxN / 2) * (kMaxN / 2)], elast[kMaxN / 2], emid[kMaxN / 2 + 1];
    int eo_e;
    T w[kMaxN];          // Gauss weights on [0,1]
    T b[2][kMaxN], g[2][kMaxN];   // Gauss-point Lagrange basis at x = 0 / 1: value, derivative
    T fb[2][kMaxN], fg[2][kMaxN]; // element basis at x = 0 / 1: value, derivative
    T K[6];              // det J * J^-1 J^-T: xx yy zz xy xz yz
    T cn[3][4];          // cn[d][a] = n_d . grad xi_a, n_d the unit normal towards +xi_d
    T fw[4];             // face JxW without the quadrature weight, per direction
    T sigma[4];          // penalty (p+1)^2 |n_d . grad xi_d|
    T hderiv;            // derivative of the first Hermite-like function at x = 0
  };

  template <typename T>
  struct DGArgs
  {
    const T          *src;
    const T          *rhs;
    T                *dst;
    const int32_t    *neigh;
    const DGConst<T> *c;
    const T          *inv_diag; // [64][(p+1)^3]
    const uint32_t   *cell_list; // cells of this launch (nullptr: cells cell_first ... in order)
    uint32_t          cell_first;
    uint32_t          n_cells;   // cells of this launch
    uint32_t          n_owned;   // owned cells of the operator: neighbour entries >= n_owned are ghosts
    T                 f1, f2;
    int               iteration_index;
    // cells of a launch without a list: cell_first + cell_stride * i
    uint32_t          cell_stride;
    // kCgSums: [gridDim.x][4] block sums.  kRestrict: the FE_Q vector the transformed residual is added into,
    // the compressed index table of the FE_Q cells (in the order of the DG cells), the 1D change of basis
    // [n][n] and whether the cells of the launch share no FE_Q DoF (plain adds instead of atomics)
    double           *partials;
    T                *cg;
    const uint32_t   *idx27;
    const T          *P1;
    int               plain;
  };

  enum Action
  {
    kVmult     = 0,
    kRestrict  = 1, // residual, changed to the FE_Q basis and added into the FE_Q vector
    kCgSums    = 2, // product stored, the four sums of the merged CG iteration
    kChebyshev = 3, // numbering of laplace_operator_dg.h:957-962
    kResidual  = 4,
    kJacobi    = 5  // P^-1 only (scaled by f2)
  };

  template <int P, typename T>
  struct DGCfg
  {
    static constexpr int N    = P + 1;
    static constexpr int NN2  = N * N;
    static constexpr int N3   = N * N * N;
    static constexpr int PX   = (N % 2 == 0) ? N + 1 : N; // odd row pitch: conflict-free x-lines
    static constexpr int VOL  = N * N * PX;
    static constexpr int FS   = N * PX;
    // per-cell LDS stride: congruent to the threads per cell modulo the 32 banks (of 4 B for fp32,
    // of 8 B for fp64 accesses), so that accesses of the form "thread index + constant" (z-lines,
    // face points) of neighbouring cells in one half-wave fall on consecutive banks
    static constexpr int CELL0 = 3 * VOL + 6 * FS;
    static constexpr int CELLP = CELL0 + (((NN2 - CELL0) % 32) + 32) % 32;
#ifndef MGX_DG_WG_THREADS
#define MGX_DG_WG_THREADS 128 // measured: 128-thread workgroups 4-11 % faster than 256 (barriers span two waves)
#endif
    // fp32, higher degrees: the size that fills its lanes best (p = 8: one cell = 81 of 128 lanes, three
    // cells = 243 of 256).  Measured on the merged Chebyshev step against 128 threads: p = 5 192 threads +3 %
    // (256: -4 %), p = 6 256 threads +7 % (192: -7 %), p = 8 256 threads +13 % (192: +9 %), p = 9 256: -11 %.
    static constexpr int WG    = sizeof(T) != 4 ? MGX_DG_WG_THREADS : (P == 5 ? 192 : (P == 6 || P == 8 ? 256 : MGX_DG_WG_THREADS));
    static constexpr int CPW_T = (WG / NN2) > 0 ? WG / NN2 : 1;
    static constexpr int CPW_L = 65536 / (CELLP * (int)sizeof(T));
    static constexpr int CPW   = CPW_T < CPW_L ? CPW_T : CPW_L;
    // ... unless the padding costs a workgroup per CU (p = 3: 13 instead of 14; measured +4.7 % without it,
    // while p = 4 and 5, where the count stays, are 1 % faster with it)
    static constexpr int CELL = (163840 / (CPW * CELL0 * (int)sizeof(T)) > 163840 / (CPW * CELLP * (int)sizeof(T))) ? CELL0 : CELLP;
    static constexpr int THREADS = ((CPW * NN2 + 63) / 64) * 64;
    // Waves per SIMD the register allocation is asked to allow (fp32; the fp64 kernels are bound by their
    // LDS at 3 waves).  One more than the compiler takes by itself where the LDS admits it and the cut is
    // below ~10 registers -- measured on the merged Chebyshev step: p = 4 76 -> 72 VGPRs, 6 -> 7 waves,
    // +3...7 % (8 waves = 64 VGPRs spill: -10 %); p = 5 89 -> 80, 5 -> 6 waves, +8...11 %; p = 6 +1...2 %;
    // p = 8 112 -> 96, 4 -> 5 waves, +7...8 %.
    static constexpr int MINW = sizeof(T) != 4 ? 1 : (P == 4 ? 7 : (P == 5 || P == 6 ? 6 : (P == 8 ? 5 : 1)));
    static_assert(CPW >= 1, "cell does not fit the LDS");
  };

  template <int N, typename T>
  __device__ __forceinline__ void ld_line(const T *a, int base, int stride, T (&r)[N])
  {
#pragma unroll
    for (int q = 0; q < N; ++q)
      r[q] = a[base + q * stride];
  }

  template <int N, typename T>
  __device__ __forceinline__ void st_line(T *a, int base, int stride, const T (&r)[N])
  {
#pragma unroll
    for (int q = 0; q < N; ++q)
      a[base + q * stride] = r[q];
  }

  // out[i] = sum_q M[q*N+i] in[q]   (M wave-uniform: scalar loads).  Two neighbouring outputs share
  // the input value and take two consecutive matrix entries: written on 2-vectors so that fp32
  // becomes v_pk_fma_f32 with the matrix pair in a scalar register pair.
  template <int N, typename T>
  __device__ __forceinline__ void mul_t(const T *__restrict__ M, const T (&in)[N], T (&out)[N])
  {
    typedef T T2 __attribute__((ext_vector_type(2)));
    constexpr int H = N / 2;
    T2            acc[H > 0 ? H : 1];
    T             last = 0;
#pragma unroll
    for (int q = 0; q < N; ++q)
      {
        const T2 x = {in[q], in[q]};
#pragma unroll
        for (int h = 0; h < H; ++h)
          {
            const T2 m = {M[q * N + 2 * h], M[q * N + 2 * h + 1]};
            acc[h]     = q == 0 ? m * x : __builtin_elementwise_fma(m, x, acc[h]);
          }
        if (N % 2)
          last = q == 0 ? M[N - 1] * in[0] : fma(M[q * N + N - 1], in[q], last);
      }
#pragma unroll
    for (int h = 0; h < H; ++h)
      {
        out[2 * h]     = acc[h][0];
        out[2 * h + 1] = acc[h][1];
      }
    if (N % 2)
      out[N - 1] = last;
  }

  // out[q] = sum_i M[q*N+i] in[i], given the transposed matrix Mt[i*N+q] = M[q*N+i]
  template <int N, typename T>
  __device__ __forceinline__ void mul(const T *__restrict__ Mt, const T (&in)[N], T (&out)[N])
  {
    mul_t<N, T>(Mt, in, out);
  }

  // The same product, out[i] = sum_q M[q][i] in[q], for a matrix with the reversal symmetry of sign SIGN
  // in even-odd form (the reference's apply_1d_matvec_kernel, matrix_vector_kernel.h:47-113): the input
  // is split into xe = in[q] + in[n-1-q] and xo = in[q] - in[n-1-q]; u = ce^T xe and v = co^T xo are two
  // independent half-size products (one 2-vector FMA per entry: v_pk_fma_f32 in fp32), out[i] = u + v,
  // out[n-1-i] = SIGN (u - v).  n^2 multiply-adds become n^2 / 2 + n additions.
  template <int N, typename T, int SIGN>
  __device__ __forceinline__ void mul_eo(const EOLine<T> &A, const T (&in)[N], T (&out)[N])
  {
    typedef T T2 __attribute__((ext_vector_type(2)));
    constexpr int H = N / 2;
    T2            x[H > 0 ? H : 1];
#pragma unroll
    for (int q = 0; q < H; ++q)
      x[q] = T2{in[q] + in[N - 1 - q], in[q] - in[N - 1 - q]};
#pragma unroll
    for (int i = 0; i < H; ++i)
      {
        T2 acc = T2{A.eo[2 * i], A.eo[2 * i + 1]} * x[0];
#pragma unroll
        for (int q = 1; q < H; ++q)
          acc = __builtin_elementwise_fma(T2{A.eo[2 * (q * H + i)], A.eo[2 * (q * H + i) + 1]}, x[q], acc);
        if (N % 2)
          acc[0] = fma(A.mrow[i], in[H], acc[0]);
        out[i]         = acc[0] + acc[1];
        out[N - 1 - i] = SIGN > 0 ? acc[0] - acc[1] : acc[1] - acc[0];
      }
    if (N % 2)
      {
        T m = SIGN > 0 ? A.mm * in[H] : T(0);
#pragma unroll
        for (int q = 0; q < H; ++q)
          m = fma(A.mcol[q], SIGN > 0 ? x[q][0] : x[q][1], m);
        out[H] = m;
      }
  }

  // which form a (degree, number type) instantiation uses.  In fp32 the dense product already runs on
  // 2-vectors (13 packed FMAs per line at p = 4 against 12 instructions plus the additions of the
  // even-odd one): the gain starts small and grows with the degree; fp64 has no packed FMA.
  template <int N, typename T>
  struct LineForm
  {
#ifdef MGX_DG_EVEN_ODD
    static constexpr bool eo = MGX_DG_EVEN_ODD != 0;
#else
    static constexpr bool eo = N >= 5; // measured (merged Chebyshev step, MI355X): fp32 p = 3 -2 %, p = 4 +4 %, p = 6 +32 %, p = 8 +60 %; fp64 p = 4 +13 %, p = 8 +118 %
#endif
  };
  // the four sweeps of the kernel: values S / S^T (symmetric), derivative D / D^T (antisymmetric);
  // in (mul_t convention) out[i] = sum_q M[q][i] in[q]
  template <int N, typename T>
  __device__ __forceinline__ void mul_S(const DGConst<T> *__restrict__ c, const T (&in)[N], T (&out)[N]) // mul_t(c->S)
  {
    if constexpr (LineForm<N, T>::eo)
      mul_eo<N, T, 1>(c->eoS, in, out);
    else
      mul_t<N, T>(c->S, in, out);
  }
  template <int N, typename T>
  __device__ __forceinline__ void mul_St(const DGConst<T> *__restrict__ c, const T (&in)[N], T (&out)[N]) // mul(c->St)
  {
    if constexpr (LineForm<N, T>::eo)
      mul_eo<N, T, 1>(c->eoSt, in, out);
    else
      mul_t<N, T>(c->St, in, out);
  }
  template <int N, typename T>
  __device__ __forceinline__ void mul_D(const DGConst<T> *__restrict__ c, const T (&in)[N], T (&out)[N]) // mul_t(c->D)
  {
    if constexpr (LineForm<N, T>::eo)
      mul_eo<N, T, -1>(c->eoD, in, out);
    else
      mul_t<N, T>(c->D, in, out);
  }
  template <int N, typename T>
  __device__ __forceinline__ void mul_Dt(const DGConst<T> *__restrict__ c, const T (&in)[N], T (&out)[N]) // mul(c->Dt)
  {
    if constexpr (LineForm<N, T>::eo)
      mul_eo<N, T, -1>(c->eoDt, in, out);
    else
      mul_t<N, T>(c->Dt, in, out);
  }

  template <int N, typename T>
  __device__ __forceinline__ T dot_line(const T *__restrict__ v, const T (&in)[N])
  {
    T s = v[0] * in[0];
#pragma unroll
    for (int i = 1; i < N; ++i)
      s += v[i] * in[i];
    return s;
  }

  template <int N, typename T>
  __device__ __forceinline__ void copy_line(const T (&in)[N], T (&out)[N])
  {
#pragma unroll
    for (int i = 0; i < N; ++i)
      out[i] = in[i];
  }

  // q[e] = sum_i E[i][e] r[i] (to the eigenvector basis) and r[i] = sum_e E[i][e] q[e] (back): every
  // eigenvector is even or odd, so the even ones see r[i] + r[n-1-i] only, the odd ones r[i] - r[n-1-i]
  template <int N, typename T>
  __device__ __forceinline__ void mul_E(const DGConst<T> *__restrict__ c, const T (&r)[N], T (&q)[N])
  {
    typedef T T2 __attribute__((ext_vector_type(2)));
    constexpr int H = N / 2, Ne = N - H, No = H;
    if constexpr (!LineForm<N, T>::eo)
      return mul_t<N, T>(c->E, r, q);
    T2 x[H > 0 ? H : 1];
#pragma unroll
    for (int i = 0; i < H; ++i)
      x[i] = T2{r[i] + r[N - 1 - i], r[i] - r[N - 1 - i]};
#pragma unroll
    for (int k = 0; k < No; ++k)
      {
        T2 acc = T2{c->epair[2 * k], c->epair[2 * k + 1]} * x[0];
#pragma unroll
        for (int i = 1; i < H; ++i)
          acc = __builtin_elementwise_fma(T2{c->epair[2 * (i * No + k)], c->epair[2 * (i * No + k) + 1]}, x[i], acc);
        if (N % 2)
          acc[0] = fma(c->emid[k], r[H], acc[0]);
        q[k]      = acc[0];
        q[Ne + k] = acc[1];
      }
    if (N % 2)
      {
        T m = c->emid[Ne - 1] * r[H];
#pragma unroll
        for (int i = 0; i < H; ++i)
          m = fma(c->elast[i], x[i][0], m);
        q[Ne - 1] = m;
      }
  }
  template <int N, typename T>
  __device__ __forceinline__ void mul_Et(const DGConst<T> *__restrict__ c, const T (&q)[N], T (&r)[N])
  {
    typedef T T2 __attribute__((ext_vector_type(2)));
    constexpr int H = N / 2, Ne = N - H, No = H;
    if constexpr (!LineForm<N, T>::eo)
      return mul_t<N, T>(c->Et, q, r);
#pragma unroll
    for (int i = 0; i < H; ++i)
      {
        T2 acc = T2{c->epair[2 * (i * No)], c->epair[2 * (i * No) + 1]} * T2{q[0], q[Ne]};
#pragma unroll
        for (int k = 1; k < No; ++k)
          acc = __builtin_elementwise_fma(T2{c->epair[2 * (i * No + k)], c->epair[2 * (i * No + k) + 1]}, T2{q[k], q[Ne + k]}, acc);
        if (N % 2)
          acc[0] = fma(c->elast[i], q[Ne - 1], acc[0]);
        r[i]         = acc[0] + acc[1];
        r[N - 1 - i] = acc[0] - acc[1];
      }
    if (N % 2)
      {
        T m = c->emid[0] * q[0];
#pragma unroll
        for (int e = 1; e < Ne; ++e)
          m = fma(c->emid[e], q[e], m);
        r[H] = m;
      }
  }

  // block-Jacobi in the eigenvector basis on the x-lines held in registers:  r <- T D^-1 T^T r
  // (JacobiTransformed::do_local_operation, laplace_operator_dg.h:2086-2097).  U is scratch.
  template <int P, typename T>
  __device__ __forceinline__ void jacobi_local(const DGConst<T> *__restrict__ c, const T *__restrict__ inv_diag, T *U,
                                               bool active, int a, int b, T (&r)[P + 1])
  {
    using C         = DGCfg<P, T>;
    constexpr int N = C::N, PX = C::PX;
    T             q[N];
    if (active)
      {
        mul_E<N>(c, r, q); // out[e] = sum_i E[i][e] r[i]
        st_line<N>(U, (b * N + a) * PX, 1, q);
      }
    __syncthreads();
    if (active)
      {
        ld_line<N>(U, b * N * PX + a, PX, r);
        mul_E<N>(c, r, q);
        st_line<N>(U, b * N * PX + a, PX, q);
      }
    __syncthreads();
    if (active)
      {
        ld_line<N>(U, b * PX + a, N * PX, r);
        mul_E<N>(c, r, q);
#pragma unroll
        for (int k = 0; k < N; ++k)
          q[k] *= inv_diag[(k * N + b) * N + a];
        mul_Et<N>(c, q, r); // out[i] = sum_e E[i][e] q[e]
        st_line<N>(U, b * PX + a, N * PX, r);
      }
    __syncthreads();
    if (active)
      {
        ld_line<N>(U, b * N * PX + a, PX, q);
        mul_Et<N>(c, q, r);
        st_line<N>(U, b * N * PX + a, PX, r);
      }
    __syncthreads();
    if (active)
      {
        ld_line<N>(U, (b * N + a) * PX, 1, q);
        mul_Et<N>(c, q, r);
      }
  }

  // residual x-lines in registers -> coefficients of the FE_Q basis of the cell: r <- (P1 x P1 x P1)^T r
  // (the transposed embedding, laplace_operator_dg.h:1803 local_basis_transformer->apply<true>).  U is scratch.
  template <int P, typename T>
  __device__ __forceinline__ void to_fe_q_local(const T *__restrict__ P1, T *U, bool active, int a, int b, T (&r)[P + 1])
  {
    using C         = DGCfg<P, T>;
    constexpr int N = C::N, PX = C::PX;
    T             q[N];
    auto mulT = [&](const T(&in)[N], T(&out)[N]) { // out[m] = sum_i P1[i][m] in[i]
#pragma unroll
      for (int m = 0; m < N; ++m)
        {
          T s = P1[m] * in[0];
#pragma unroll
          for (int i = 1; i < N; ++i)
            s = fma(P1[i * N + m], in[i], s);
          out[m] = s;
        }
    };
    if (active)
      {
        mulT(r, q);
        st_line<N>(U, (b * N + a) * PX, 1, q);
      }
    __syncthreads();
    if (active)
      {
        ld_line<N>(U, b * N * PX + a, PX, r);
        mulT(r, q);
        st_line<N>(U, b * N * PX + a, PX, q);
      }
    __syncthreads();
    if (active)
      {
        ld_line<N>(U, b * PX + a, N * PX, r);
        mulT(r, q);
        st_line<N>(U, b * PX + a, N * PX, q);
      }
    __syncthreads();
    if (active)
      ld_line<N>(U, (b * N + a) * PX, 1, r);
  }

  // r[0 .. p] of the x-line (j, k) of an FE_Q cell added into the vector through the compressed index table
  // (27 entities per cell: first DoF of every vertex / line / quad / hex entity, vector_access_reduced.h:153-247)
  template <int P, typename T>
  __device__ __forceinline__ void add_fe_q_line(T *__restrict__ dst, const uint32_t *__restrict__ idx27, uint32_t cell, int j,
                                                int k, const T (&r)[P + 1], bool plain)
  {
    const int       cy = j == 0 ? 0 : (j == P ? 2 : 1), cz = k == 0 ? 0 : (k == P ? 2 : 1);
    const int       oy = cy == 1 ? j - 1 : 0, oz = cz == 1 ? k - 1 : 0;
    const uint32_t *ind = idx27 + 27u * (size_t)cell + 3 * (3 * cz + cy);
    const uint32_t  off = (uint32_t)((cy == 1 ? P - 1 : 1) * oz + oy);
    const uint32_t  b0 = ind[0], b1 = ind[1], b2 = ind[2];
    auto add = [&](uint32_t at, T v) {
      if (plain)
        dst[at] += v;
      else
        unsafeAtomicAdd(&dst[at], v);
    };
    if (b0 != 0xFFFFFFFFu)
      add(b0 + off, r[0]);
    if (b1 != 0xFFFFFFFFu)
      {
#pragma unroll
        for (int i = 0; i < P - 1; ++i)
          add(b1 + off * (uint32_t)(P - 1) + (uint32_t)i, r[1 + i]);
      }
    if (b2 != 0xFFFFFFFFu)
      add(b2 + off, r[P]);
  }

  // GHOSTS (Hermite-like basis on a decomposed mesh): neighbour entries >= A.n_owned are ghost faces
  template <int P, typename T, int TYPE, int ACTION, bool GHOSTS = false>
  __global__ void __launch_bounds__((DGCfg<P, T>::THREADS), (DGCfg<P, T>::MINW)) dg_cell_kernel(const DGArgs<T> A)
  {
    using C         = DGCfg<P, T>;
    constexpr int N = C::N, NN2 = C::NN2, N3 = C::N3, PX = C::PX, VOL = C::VOL, FS = C::FS;
    __shared__ __attribute__((aligned(16))) T lds[C::CPW * C::CELL];
#ifdef MGX_DG_LDS_PAD // occupancy experiment (tools/experiments): extra LDS per workgroup, in bytes
    __shared__ char lds_pad[MGX_DG_LDS_PAD];
    if (A.n_cells == 0xFFFFFFFFu) // never true; keeps the array allocated
      {
        lds_pad[threadIdx.x] = 1;
        __syncthreads();
        A.dst[0] = (T)lds_pad[(threadIdx.x + 1) % MGX_DG_LDS_PAD];
      }
#endif

    const DGConst<T> *__restrict__ c = A.c;
    const int  tid    = threadIdx.x;
    const int  cw     = tid / NN2;
    const int  t      = tid - cw * NN2;
    const int  a      = t % N, b = t / N; // line owner (a, b) = face point (a, b)
    const bool active = cw < C::CPW;
    // XCD-aware block order: the dispatcher deals workgroups round-robin over the 8 XCDs (each with its
    // own L2); give every XCD one contiguous eighth of the cells, which lie along a space-filling curve,
    // so that most face neighbours are read through the L2 that holds them (measured: vmult +1...4 %,
    // the VALU-bound merged Chebyshev step +0.3 %)
    const uint32_t xq = gridDim.x / 8, xr = gridDim.x % 8, xcd = blockIdx.x % 8;
    const uint32_t bid = xcd * xq + (xcd < xr ? xcd : xr) + blockIdx.x / 8;
    uint32_t   cell   = bid * C::CPW + (active ? cw : 0);
    const bool store  = active && cell < A.n_cells;
    if (cell >= A.n_cells)
      cell = A.n_cells - 1;
    cell = A.cell_list ? A.cell_list[cell] : cell * A.cell_stride + A.cell_first;

    T *U  = lds + (active ? cw : 0) * C::CELL;
    T *GY = U + VOL, *GZ = U + 2 * VOL;
    T *F  = U + 3 * VOL; // face scratch of the direction in work: [2 faces][3][FS]
    const int fidx = b * PX + a;

    const T *__restrict__ src = A.src;
    const size_t cbase = (size_t)cell * N3;
    T            xs[N]; // the source x-line: needed again by the Chebyshev update
    int          nb[6];
    unsigned     cat = 0;
#pragma unroll
    for (int f = 0; f < 6; ++f)
      {
        nb[f] = A.neigh[(size_t)cell * 6 + f];
        cat |= (nb[f] < 0 ? 1u : 0u) << f;
      }

    if constexpr (ACTION == kJacobi)
      {
        T r[N];
        if (active)
          {
#pragma unroll
            for (int i = 0; i < N; ++i)
              r[i] = src[cbase + (b * N + a) * N + i];
          }
        jacobi_local<P, T>(c, A.inv_diag + (size_t)cat * N3, U, active, a, b, r);
        if (store)
          {
#pragma unroll
            for (int i = 0; i < N; ++i)
              A.dst[cbase + (b * N + a) * N + i] = A.f2 * r[i];
          }
        return;
      }

    // own traces (value, reference normal derivative) of the 6 faces at this thread's face point,
    // and what the faces give back to the integration (value / normal-derivative test function):
    // registers -- the line owner (a, b) of a sweep direction is the owner of face point (a, b)
    T To[6], No[6], Vf[6], Wf[6];

    // ---- 1. source x-line -> Gauss values along x
    if (active)
      {
        T u[N];
#pragma unroll
        for (int i = 0; i < N; ++i)
          xs[i] = src[cbase + (b * N + a) * N + i];
        if constexpr (TYPE != MGX_DG_GAUSS)
          mul_St<N>(c, xs, u);
        else
          copy_line<N>(xs, u);
        st_line<N>(U, (b * N + a) * PX, 1, u);
      }
    __syncthreads();

    // ---- 2. Gauss values along y
    if constexpr (TYPE != MGX_DG_GAUSS)
      {
        if (active)
          {
            T u[N], v[N];
            ld_line<N>(U, b * N * PX + a, PX, u);
            mul_St<N>(c, u, v);
            st_line<N>(U, b * N * PX + a, PX, v);
          }
        __syncthreads();
      }

    // ---- 3. z-lines: Gauss values along z, z-derivative, traces on the z faces
    if (active)
      {
        T u[N], v[N];
        ld_line<N>(U, b * PX + a, N * PX, u);
        if constexpr (TYPE != MGX_DG_GAUSS)
          {
            mul_St<N>(c, u, v);
            st_line<N>(U, b * PX + a, N * PX, v);
          }
        else
          copy_line<N>(u, v);
        mul_Dt<N>(c, v, u);
        st_line<N>(GZ, b * PX + a, N * PX, u);
        To[4] = dot_line<N>(c->b[0], v);
        To[5] = dot_line<N>(c->b[1], v);
        No[4] = dot_line<N>(c->g[0], v);
        No[5] = dot_line<N>(c->g[1], v);
      }
    __syncthreads();

    // ---- 4. y-lines: y-derivative and traces on the y faces; x-lines: traces on the x faces
    if (active)
      {
        T u[N], v[N];
        ld_line<N>(U, b * N * PX + a, PX, u);
        mul_Dt<N>(c, u, v);
        st_line<N>(GY, b * N * PX + a, PX, v);
        To[2] = dot_line<N>(c->b[0], u);
        To[3] = dot_line<N>(c->b[1], u);
        No[2] = dot_line<N>(c->g[0], u);
        No[3] = dot_line<N>(c->g[1], u);
        ld_line<N>(U, (b * N + a) * PX, 1, u);
        To[0] = dot_line<N>(c->b[0], u);
        To[1] = dot_line<N>(c->b[1], u);
        No[0] = dot_line<N>(c->g[0], u);
        No[1] = dot_line<N>(c->g[1], u);
      }

    // ---- 5. faces, one direction at a time (two faces): per face three arrays of (p+1)^2 words --
    // neighbour trace, then sum of the traces | neighbour normal derivative, then weighted jump |
    // tangential part of the result.  A Dirichlet face mirrors the own values (:1568-1577).
#pragma unroll
    for (int d = 0; d < 3; ++d)
      {
        const int sd = d == 0 ? 1 : (d == 1 ? N : N * N); // stride of the normal direction in a cell
        const int s1 = d == 0 ? N : 1;                    // ... of the two tangential ones (ascending)
        const int s2 = d == 2 ? N : N * N;
        const int t1 = d == 0 ? 1 : 0, t2 = d == 2 ? 1 : 2;
        auto      E0 = [&](int s) { return F + (3 * s) * FS; };
        auto      E1 = [&](int s) { return F + (3 * s + 1) * FS; };
        auto      AT = [&](int s) { return F + (3 * s + 2) * FS; };
        if (active)
          {
#pragma unroll
            for (int s = 0; s < 2; ++s)
              {
                const int f  = 2 * d + s;
                T         ev = 0, ed = 0;
                if (nb[f] >= 0)
                  {
                    const T *__restrict__ xn = src + (size_t)nb[f] * N3 + a * s1 + b * s2;
                    if constexpr (TYPE == MGX_DG_HERMITE)
                      {
                        // owned neighbour: its two node layers next to the face (its upper face for our
                        // lower one and vice versa); ghost: the (value, normal derivative) pair of the
                        // face point as its owner computed it (k_pack_faces; laplace_operator_dg.h:1015-1039
                        // sends the same pair).  Two loads either way, no divergent branch.
                        const int l0 = s == 0 ? N - 1 : 0, l1 = s == 0 ? (N > 1 ? N - 2 : 0) : (N > 1 ? 1 : 0);
                        if constexpr (GHOSTS)
                          {
                            // 32-bit entry offsets (a vector holds fewer than 2^32 entries): one select
                            const uint32_t nbu   = (uint32_t)nb[f];
                            const bool     ghost = nbu >= A.n_owned;
                            const uint32_t base  = ghost ? A.n_owned * (uint32_t)N3 + ((nbu - A.n_owned) * NN2 + b * N + a) * 2
                                                         : nbu * (uint32_t)N3 + a * s1 + b * s2;
                            const T v0 = src[base + (ghost ? 0 : l0 * sd)], v1 = src[base + (ghost ? 1 : l1 * sd)];
                            ev         = v0;
                            ed         = ghost ? v1 : (s == 0 ? c->hderiv * (v1 - v0) : c->hderiv * (v0 - v1));
                          }
                        else
                          {
                            const T v0 = xn[l0 * sd], v1 = xn[l1 * sd];
                            ev         = v0;
                            ed         = s == 0 ? c->hderiv * (v1 - v0) : c->hderiv * (v0 - v1);
                          }
                      }
                    else
                      {
                        T line[N];
#pragma unroll
                        for (int i = 0; i < N; ++i)
                          line[i] = xn[i * sd];
                        ev = dot_line<N>(c->fb[1 - s], line);
                        ed = dot_line<N>(c->fg[1 - s], line);
                      }
                  }
                E0(s)[fidx] = ev;
                E1(s)[fidx] = ed;
              }
          }
        __syncthreads();
        if constexpr (TYPE != MGX_DG_GAUSS)
          {
            // neighbour traces: in-face change to the Gauss points, first then second direction
            if (active)
              for (int L = t; L < 4 * N; L += NN2)
                {
                  T  u[N], v[N];
                  T *arr = F + ((L / N) / 2 * 3 + (L / N) % 2) * FS + (L % N) * PX;
                  ld_line<N>(arr, 0, 1, u);
                  mul_St<N>(c, u, v);
                  st_line<N>(arr, 0, 1, v);
                }
            __syncthreads();
            if (active)
              for (int L = t; L < 4 * N; L += NN2)
                {
                  T  u[N], v[N];
                  T *arr = F + ((L / N) / 2 * 3 + (L / N) % 2) * FS + (L % N);
                  ld_line<N>(arr, 0, PX, u);
                  mul_St<N>(c, u, v);
                  st_line<N>(arr, 0, PX, v);
                }
            __syncthreads();
          }
        // in the quadrature point: sum of the traces -> E0, weighted jump -> E1; sum of the normal
        // derivatives and the weighted jump stay in registers
        T sN[2], wJ[2];
        if (active)
          {
            const T wq = c->w[a] * c->w[b] * c->fw[d];
#pragma unroll
            for (int s = 0; s < 2; ++s)
              {
                const int  f = 2 * d + s;
                const bool dirichlet = nb[f] < 0;
                const T    te = dirichlet ? To[f] : E0(s)[fidx];
                const T    ne = dirichlet ? No[f] : E1(s)[fidx];
                sN[s]         = No[f] + ne;
                wJ[s]         = wq * (dirichlet ? T(2) * To[f] : To[f] - te);
                E0(s)[fidx]   = To[f] + te;
                E1(s)[fidx]   = wJ[s];
              }
          }
        __syncthreads();
        // tangential part, lines of the first tangential direction (s = +-1 the side of the face):
        //   AT = -s/2 c_t1 (w d_t1 sumT + d_t1^T wj)
        if (active)
          for (int L = t; L < 2 * N; L += NN2)
            {
              const int s = L / N, l = L % N;
              const T   half_s = s ? T(0.5) : T(-0.5);
              const T   ct = c->cn[d][t1];
              T         st[N], wj[N], ds[N], dj[N];
              ld_line<N>(E0(s), l * PX, 1, st);
              ld_line<N>(E1(s), l * PX, 1, wj);
              mul_Dt<N>(c, st, ds);
              mul_D<N>(c, wj, dj);
              const T wl = c->w[l] * c->fw[d];
#pragma unroll
              for (int i = 0; i < N; ++i)
                ds[i] = -half_s * ct * (c->w[i] * wl * ds[i] + dj[i]);
              st_line<N>(AT(s), l * PX, 1, ds);
            }
        __syncthreads();
        if (active)
          for (int L = t; L < 2 * N; L += NN2)
            {
              const int s = L / N, l = L % N;
              const T   half_s = s ? T(0.5) : T(-0.5);
              const T   ct = c->cn[d][t2];
              T         st[N], wj[N], v[N], ds[N], dj[N];
              ld_line<N>(E0(s), l, PX, st);
              ld_line<N>(E1(s), l, PX, wj);
              ld_line<N>(AT(s), l, PX, v);
              mul_Dt<N>(c, st, ds);
              mul_D<N>(c, wj, dj);
              const T wl = c->w[l] * c->fw[d];
#pragma unroll
              for (int i = 0; i < N; ++i)
                v[i] -= half_s * ct * (c->w[i] * wl * ds[i] + dj[i]);
              st_line<N>(AT(s), l, PX, v);
            }
        __syncthreads();
        // value test function  V = sigma wj - s/2 w c_n sumN + tangential part,
        // normal derivative test function  W = -s/2 c_n wj
        if (active)
          {
            const T wq = c->w[a] * c->w[b] * c->fw[d];
#pragma unroll
            for (int s = 0; s < 2; ++s)
              {
                const int f      = 2 * d + s;
                const T   half_s = s ? T(0.5) : T(-0.5);
                Vf[f] = c->sigma[d] * wJ[s] - half_s * wq * c->cn[d][d] * sN[s] + AT(s)[fidx];
                Wf[f] = -half_s * c->cn[d][d] * wJ[s];
              }
          }
      }

    // face contributions to a line's integration
    auto add_faces = [&](int d, T(&o)[N]) {
#pragma unroll
      for (int i = 0; i < N; ++i)
        o[i] += c->b[0][i] * Vf[2 * d] + c->b[1][i] * Vf[2 * d + 1] + c->g[0][i] * Wf[2 * d] + c->g[1][i] * Wf[2 * d + 1];
    };
    __syncthreads(); // GY complete; the face scratch is not touched below

    // ---- 6. x-lines: gradient, coefficient (laplace_operator_dg.h:1700-1716), integration along x
    if (active)
      {
        T u[N], gx[N], gy[N], gz[N], o[N];
        ld_line<N>(U, (b * N + a) * PX, 1, u);
        mul_Dt<N>(c, u, gx);
        ld_line<N>(GY, (b * N + a) * PX, 1, gy);
        ld_line<N>(GZ, (b * N + a) * PX, 1, gz);
        const T wab = c->w[a] * c->w[b];
#pragma unroll
        for (int i = 0; i < N; ++i)
          {
            const T wq = wab * c->w[i];
            const T fx = wq * (c->K[0] * gx[i] + c->K[3] * gy[i] + c->K[4] * gz[i]);
            const T fy = wq * (c->K[3] * gx[i] + c->K[1] * gy[i] + c->K[5] * gz[i]);
            const T fz = wq * (c->K[4] * gx[i] + c->K[5] * gy[i] + c->K[2] * gz[i]);
            gx[i]      = fx;
            gy[i]      = fy;
            gz[i]      = fz;
          }
        st_line<N>(GY, (b * N + a) * PX, 1, gy);
        st_line<N>(GZ, (b * N + a) * PX, 1, gz);
        mul_D<N>(c, gx, o);
        add_faces(0, o);
        st_line<N>(U, (b * N + a) * PX, 1, o);
      }
    __syncthreads();
    // ---- 7. y-lines
    if (active)
      {
        T fy[N], o[N], u[N];
        ld_line<N>(GY, b * N * PX + a, PX, fy);
        ld_line<N>(U, b * N * PX + a, PX, u);
        mul_D<N>(c, fy, o);
        add_faces(1, o);
#pragma unroll
        for (int i = 0; i < N; ++i)
          o[i] += u[i];
        st_line<N>(U, b * N * PX + a, PX, o);
      }
    __syncthreads();
    // ---- 8. z-lines, then back to the element basis along z
    if (active)
      {
        T fz[N], o[N], u[N];
        ld_line<N>(GZ, b * PX + a, N * PX, fz);
        ld_line<N>(U, b * PX + a, N * PX, u);
        mul_D<N>(c, fz, o);
        add_faces(2, o);
#pragma unroll
        for (int i = 0; i < N; ++i)
          o[i] += u[i];
        if constexpr (TYPE != MGX_DG_GAUSS)
          {
            mul_S<N>(c, o, u);
            st_line<N>(U, b * PX + a, N * PX, u);
          }
        else
          st_line<N>(U, b * PX + a, N * PX, o);
      }
    __syncthreads();
    if constexpr (TYPE != MGX_DG_GAUSS)
      {
        if (active)
          {
            T u[N], v[N];
            ld_line<N>(U, b * N * PX + a, PX, u);
            mul_S<N>(c, u, v);
            st_line<N>(U, b * N * PX + a, PX, v);
          }
        __syncthreads();
      }
    // ---- 10. x-lines: result in the element basis, epilogue of the action
    T y[N];
    if (active)
      {
        T u[N];
        ld_line<N>(U, (b * N + a) * PX, 1, u);
        if constexpr (TYPE != MGX_DG_GAUSS)
          mul_S<N>(c, u, y);
        else
          copy_line<N>(u, y);
      }
    const size_t lbase = cbase + (b * N + a) * N;
    if constexpr (ACTION == kVmult)
      {
        if (store)
          {
#pragma unroll
            for (int i = 0; i < N; ++i)
              A.dst[lbase + i] = y[i];
          }
      }
    else if constexpr (ACTION == kResidual)
      {
        if (store)
          {
#pragma unroll
            for (int i = 0; i < N; ++i)
              A.dst[lbase + i] = A.rhs[lbase + i] - y[i];
          }
      }
    else if constexpr (ACTION == kRestrict)
      {
        // laplace_operator_dg.h:1798-1819
        if (active)
          {
#pragma unroll
            for (int i = 0; i < N; ++i)
              y[i] = A.rhs[lbase + i] - y[i];
          }
        __syncthreads(); // the x-lines above were read from U
        to_fe_q_local<P, T>(A.P1, U, active, a, b, y);
        if (store)
          add_fe_q_line<P, T>(A.cg, A.idx27, cell, a, b, y, A.plain != 0);
      }
    else if constexpr (ACTION == kCgSums)
      {
        // laplace_operator_dg.h:1827-1838: dst.src, rhs.rhs, dst.rhs, dst.dst over the cells of the launch
        double sum[4] = {0., 0., 0., 0.};
        if (store)
          {
#pragma unroll
            for (int i = 0; i < N; ++i)
              {
                const T r        = A.rhs[lbase + i];
                A.dst[lbase + i] = y[i];
                sum[0] += (double)(y[i] * xs[i]);
                sum[1] += (double)(r * r);
                sum[2] += (double)(y[i] * r);
                sum[3] += (double)(y[i] * y[i]);
              }
          }
        constexpr int WAVES = C::THREADS / 64;
        double       *red   = reinterpret_cast<double *>(lds);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k)
          {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1)
              sum[k] += __shfl_down(sum[k], o);
            if ((tid & 63) == 0)
              red[(tid >> 6) * 4 + k] = sum[k];
          }
        __syncthreads();
        if (tid < 4)
          {
            double t4 = red[tid];
#pragma unroll
            for (int w = 1; w < WAVES; ++w)
              t4 += red[w * 4 + tid];
            A.partials[(size_t)blockIdx.x * 4 + tid] = t4;
          }
      }
    else
      {
        // laplace_operator_dg.h:1839-1860
        if (active)
          {
#pragma unroll
            for (int i = 0; i < N; ++i)
              y[i] = A.rhs[lbase + i] - y[i];
          }
        jacobi_local<P, T>(c, A.inv_diag + (size_t)cat * N3, U, active, a, b, y);
        if (store)
          {
            const T f1p = T(1) + A.f1;
            if (A.iteration_index == 1)
              {
#pragma unroll
                for (int i = 0; i < N; ++i)
                  A.dst[lbase + i] = A.f2 * y[i] + f1p * xs[i];
              }
            else
              {
#pragma unroll
                for (int i = 0; i < N; ++i)
                  A.dst[lbase + i] = A.f2 * y[i] + f1p * xs[i] - A.f1 * A.dst[lbase + i];
              }
          }
      }
  }

  // ------------------------------------------------------------------------------------------
  // host: 1D data
  struct Poly1 // c * prod (x - r_k)
  {
    double              c = 1;
    std::vector<double> r;
    double val(double x) const
    {
      double v = c;
      for (double rk : r)
        v *= x - rk;
      return v;
    }
    double der(double x) const
    {
      double s = 0;
      for (size_t m = 0; m < r.size(); ++m)
        {
          double v = c;
          for (size_t k = 0; k < r.size(); ++k)
            if (k != m)
              v *= x - r[k];
          s += v;
        }
      return s;
    }
    void normalise(double x, double value) { c *= value / val(x); }
  };

  // eigenvalues (ascending) and eigenvectors (columns of V) of a symmetric matrix, cyclic Jacobi
  void sym_eig(int n, std::vector<double> A, std::vector<double> &lambda, std::vector<double> &V)
  {
    V.assign((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i)
      V[i * n + i] = 1;
    for (int sweep = 0; sweep < 100; ++sweep)
      {
        double off = 0, dia = 0;
        for (int i = 0; i < n; ++i)
          for (int j = 0; j < n; ++j)
            (i == j ? dia : off) += A[i * n + j] * A[i * n + j];
        if (off <= 1e-30 * dia)
          break;
        for (int p = 0; p < n - 1; ++p)
          for (int q = p + 1; q < n; ++q)
            {
              if (std::abs(A[p * n + q]) < 1e-300)
                continue;
              const double theta = (A[q * n + q] - A[p * n + p]) / (2 * A[p * n + q]);
              const double tt    = (theta >= 0 ? 1.0 : -1.0) / (std::abs(theta) + std::sqrt(theta * theta + 1));
              const double cs = 1 / std::sqrt(tt * tt + 1), sn = tt * cs;
              for (int k = 0; k < n; ++k)
                {
                  const double akp = A[k * n + p], akq = A[k * n + q];
                  A[k * n + p] = cs * akp - sn * akq;
                  A[k * n + q] = sn * akp + cs * akq;
                }
              for (int k = 0; k < n; ++k)
                {
                  const double apk = A[p * n + k], aqk = A[q * n + k];
                  A[p * n + k] = cs * apk - sn * aqk;
                  A[q * n + k] = sn * apk + cs * aqk;
                }
              for (int k = 0; k < n; ++k)
                {
                  const double vkp = V[k * n + p], vkq = V[k * n + q];
                  V[k * n + p] = cs * vkp - sn * vkq;
                  V[k * n + q] = sn * vkp + cs * vkq;
                }
            }
      }
    std::vector<int> order(n);
    std::iota(order.begin(), order.end(), 0);
    std::sort(order.begin(), order.end(), [&](int i, int j) { return A[i * n + i] < A[j * n + j]; });
    lambda.resize(n);
    std::vector<double> Vs((size_t)n * n);
    for (int e = 0; e < n; ++e)
      {
        lambda[e] = A[order[e] * n + order[e]];
        for (int k = 0; k < n; ++k)
          Vs[k * n + e] = V[k * n + order[e]];
      }
    V.swap(Vs);
  }

  // roots of the Jacobi polynomial P^(al,al)_m mapped to [0,1]: eigenvalues of the Jacobi matrix
  std::vector<double> jacobi_roots01(int m, double al)
  {
    std::vector<double> out;
    if (m <= 0)
      return out;
    std::vector<double> J((size_t)m * m, 0.0), lam, V;
    for (int k = 1; k < m; ++k)
      {
        const double s = 2 * k + 2 * al;
        const double bk =
          2.0 / s * std::sqrt(k * (k + al) * (k + al) * (k + 2 * al) / ((s - 1) * (s + 1)));
        J[(k - 1) * m + k] = J[k * m + k - 1] = bk;
      }
    sym_eig(m, J, lam, V);
    for (double x : lam)
      out.push_back(0.5 * (x + 1));
    return out;
  }

  void gauss01(int n, std::vector<double> &x, std::vector<double> &w)
  {
    x = jacobi_roots01(n, 0.0);
    w.resize(n);
    for (int i = 0; i < n; ++i)
      {
        // Newton polish on the Legendre polynomial, weight 1 / ((1 - t^2) P_n'(t)^2) on [0,1]
        double t = 2 * x[i] - 1, dp = 0;
        for (int it = 0; it < 3; ++it)
          {
            double p0 = 1, p1 = t;
            for (int k = 2; k <= n; ++k)
              {
                const double pk = ((2 * k - 1) * t * p1 - (k - 1) * p0) / k;
                p0 = p1;
                p1 = pk;
              }
            if (n == 1)
              {
                p0 = 1;
                p1 = t;
              }
            dp = n * (t * p1 - p0) / (t * t - 1);
            if (it < 2)
              t -= p1 / dp;
          }
        x[i] = 0.5 * (t + 1);
        w[i] = 1.0 / ((1 - t * t) * dp * dp);
      }
  }

  std::vector<Poly1> lagrange(const std::vector<double> &nodes)
  {
    std::vector<Poly1> out(nodes.size());
    for (size_t i = 0; i < nodes.size(); ++i)
      {
        for (size_t k = 0; k < nodes.size(); ++k)
          if (k != i)
            out[i].r.push_back(nodes[k]);
        out[i].normalise(nodes[i], 1.0);
      }
    return out;
  }

  // FE_DGQHermite's 1D functions (deal.II Polynomials::HermiteLikeInterpolation, external, restated
  // from its documented construction): p_0 is the only function with a value at x = 0, p_0 and p_1
  // the only ones with a derivative there (mirror image at x = 1), p_0 is L2-orthogonal to p_1, the
  // inner functions are Lagrange polynomials in the roots of the Jacobi polynomial P^(4,4)_{p-3}
  // times x^2 (1-x)^2, and all functions sum to one (hence p_1'(0) = -p_0'(0), which
  // laplace_operator_dg.h:1190-1198 relies on).  Degree 1: hat functions, 2: Bernstein.
  std::vector<Poly1> hermite_like(int p)
  {
    std::vector<Poly1> out(p + 1);
    if (p == 0)
      return out;
    if (p == 1)
      {
        out[0].r = {1.0};
        out[0].c = -1;
        out[1].r = {0.0};
        return out;
      }
    if (p == 2)
      {
        out[0].r = {1.0, 1.0};
        out[1].r = {0.0, 1.0};
        out[1].c = -2;
        out[2].r = {0.0, 0.0};
        return out;
      }
    const std::vector<double> inner = jacobi_roots01(p - 3, 4.0);
    Poly1                     q0, q1;
    q0.r = {1.0, 1.0};
    q1.r = {0.0, 1.0, 1.0};
    for (double x : inner)
      {
        q0.r.push_back(x);
        q1.r.push_back(x);
      }
    std::vector<double> xq, wq;
    gauss01(p + 2, xq, wq); // exact to degree 2p + 3 >= deg(x q0 q1) = 2p
    double i0 = 0, i1 = 0;
    for (size_t k = 0; k < xq.size(); ++k)
      {
        i0 += wq[k] * q0.val(xq[k]) * q1.val(xq[k]);
        i1 += wq[k] * xq[k] * q0.val(xq[k]) * q1.val(xq[k]);
      }
    Poly1 p0 = q0;
    p0.r.push_back(i1 / i0);
    p0.normalise(0.0, 1.0);
    Poly1 p1 = q1;
    p1.c     = -p0.der(0.0) / q1.der(0.0);
    out[0]   = p0;
    out[1]   = p1;
    for (size_t j = 0; j < inner.size(); ++j)
      {
        Poly1 f;
        f.r = {0.0, 0.0, 1.0, 1.0};
        for (size_t k = 0; k < inner.size(); ++k)
          if (k != j)
            f.r.push_back(inner[k]);
        f.normalise(inner[j], 1.0);
        out[2 + j] = f;
      }
    auto mirror = [](const Poly1 &f) {
      Poly1 m;
      m.c = f.c * ((f.r.size() % 2) ? -1.0 : 1.0);
      for (double r : f.r)
        m.r.push_back(1.0 - r);
      return m;
    };
    out[p - 1] = mirror(p1);
    out[p]     = mirror(p0);
    return out;
  }

  struct Host1D
  {
    int                 n = 0;
    std::vector<double> xq, wq, S, SD, D, E, lambda;
    bool                e_parity = false; // eigenvectors sorted even first / odd behind (see build_1d)
    double              b[2][kMaxN], g[2][kMaxN], fb[2][kMaxN], fg[2][kMaxN];
    double              hderiv = 0;
    std::vector<double> P1; // [i*n+q]: values in the Gauss-Lobatto nodes -> coefficients of the element basis
    // eigenfunctions: Laplace form, first-derivative form, values and derivatives at the two ends
    std::vector<double> lt, ct, beta[2], gamma[2];
  };

  int build_1d(int p, int basis, Host1D &h, std::string &why)
  {
    const int n = p + 1;
    h.n         = n;
    gauss01(n, h.xq, h.wq);
    std::vector<Poly1> fe;
    if (basis == MGX_DG_HERMITE)
      fe = hermite_like(p);
    else if (basis == MGX_DG_GAUSS)
      fe = lagrange(h.xq);
    else
      {
        std::vector<double> nodes{0.0};
        for (double x : jacobi_roots01(n - 2, 1.0))
          nodes.push_back(x);
        nodes.push_back(1.0);
        fe = lagrange(nodes);
      }
    const std::vector<Poly1> col = lagrange(h.xq);
    h.S.assign(n * n, 0);
    h.SD.assign(n * n, 0);
    h.D.assign(n * n, 0);
    for (int q = 0; q < n; ++q)
      for (int i = 0; i < n; ++i)
        {
          h.S[q * n + i]  = fe[i].val(h.xq[q]);
          h.SD[q * n + i] = fe[i].der(h.xq[q]);
          h.D[q * n + i]  = col[i].der(h.xq[q]);
        }
    for (int s = 0; s < 2; ++s)
      for (int i = 0; i < n; ++i)
        {
          h.b[s][i]  = col[i].val(s);
          h.g[s][i]  = col[i].der(s);
          h.fb[s][i] = fe[i].val(s);
          h.fg[s][i] = fe[i].der(s);
        }
    h.hderiv = fe[0].der(0.0);
    {
      // embedding of FE_Q(p) (nodal in the Gauss-Lobatto points g_q) into this basis on one cell:
      // sum_i d_i phi_i(g_q) = c_q, i.e. d = B^-1 c with B[q][i] = phi_i(g_q)
      // (LocalBasisTransformer type 1, laplace_operator_dg.h:103-135, applied at :1802, :1881)
      std::vector<double> nodes{0.0};
      for (double x : jacobi_roots01(n - 2, 1.0))
        nodes.push_back(x);
      nodes.push_back(1.0);
      if (n == 1)
        nodes = {0.5};
      std::vector<double> Bm(n * n), inv(n * n, 0.0);
      for (int q = 0; q < n; ++q)
        for (int i = 0; i < n; ++i)
          Bm[q * n + i] = fe[i].val(nodes[q]);
      for (int i = 0; i < n; ++i)
        inv[i * n + i] = 1;
      for (int col = 0; col < n; ++col) // Gauss-Jordan with partial pivoting
        {
          int piv = col;
          for (int r = col + 1; r < n; ++r)
            if (std::abs(Bm[r * n + col]) > std::abs(Bm[piv * n + col]))
              piv = r;
          if (std::abs(Bm[piv * n + col]) < 1e-14)
            {
              why = "element basis is not unisolvent in the Gauss-Lobatto nodes";
              return MGX_ERR_UNSUPPORTED;
            }
          for (int k = 0; k < n; ++k)
            {
              std::swap(Bm[piv * n + k], Bm[col * n + k]);
              std::swap(inv[piv * n + k], inv[col * n + k]);
            }
          const double dinv = 1.0 / Bm[col * n + col];
          for (int k = 0; k < n; ++k)
            {
              Bm[col * n + k] *= dinv;
              inv[col * n + k] *= dinv;
            }
          for (int r = 0; r < n; ++r)
            if (r != col)
              {
                const double f = Bm[r * n + col];
                for (int k = 0; k < n; ++k)
                  {
                    Bm[r * n + k] -= f * Bm[col * n + k];
                    inv[r * n + k] -= f * inv[col * n + k];
                  }
              }
        }
      h.P1 = inv;
    }

    // generalised eigenproblem lapl v = lambda mass v (laplace_operator_dg.h:179-215)
    std::vector<double> mass(n * n, 0), lapl(n * n, 0), cfirst(n * n, 0);
    const double        pen = double(n) * n;
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j)
        {
          double m = 0, l = 0, cf = 0;
          for (int q = 0; q < n; ++q)
            {
              m += h.wq[q] * h.S[q * n + i] * h.S[q * n + j];
              l += h.wq[q] * h.SD[q * n + i] * h.SD[q * n + j];
              cf += h.wq[q] * h.S[q * n + i] * h.SD[q * n + j];
            }
          mass[i * n + j]   = m;
          cfirst[i * n + j] = cf;
          l += h.fb[0][i] * h.fb[0][j] * pen + 0.5 * (h.fg[0][i] * h.fb[0][j] + h.fg[0][j] * h.fb[0][i]);
          l += h.fb[1][i] * h.fb[1][j] * pen - 0.5 * (h.fg[1][i] * h.fb[1][j] + h.fg[1][j] * h.fb[1][i]);
          lapl[i * n + j] = l;
        }
    // Cholesky mass = L L^T
    std::vector<double> L(n * n, 0);
    for (int i = 0; i < n; ++i)
      for (int j = 0; j <= i; ++j)
        {
          double s = mass[i * n + j];
          for (int k = 0; k < j; ++k)
            s -= L[i * n + k] * L[j * n + k];
          if (i == j)
            {
              if (s <= 0)
                {
                  why = "1D mass matrix is not positive definite";
                  return MGX_ERR_UNSUPPORTED;
                }
              L[i * n + i] = std::sqrt(s);
            }
          else
            L[i * n + j] = s / L[j * n + j];
        }
    // C = L^-1 lapl L^-T
    auto solve_lower = [&](std::vector<double> &B) { // B <- L^-1 B (columns)
      for (int col_ = 0; col_ < n; ++col_)
        for (int i = 0; i < n; ++i)
          {
            double s = B[i * n + col_];
            for (int k = 0; k < i; ++k)
              s -= L[i * n + k] * B[k * n + col_];
            B[i * n + col_] = s / L[i * n + i];
          }
    };
    std::vector<double> Cm = lapl;
    solve_lower(Cm);
    std::vector<double> Ct(n * n);
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j)
        Ct[i * n + j] = Cm[j * n + i];
    solve_lower(Ct);
    for (int i = 0; i < n; ++i)
      for (int j = i + 1; j < n; ++j)
        Ct[i * n + j] = Ct[j * n + i] = 0.5 * (Ct[i * n + j] + Ct[j * n + i]);
    std::vector<double> Q;
    sym_eig(n, Ct, h.lambda, Q);
    // E = L^-T Q
    h.E.assign(n * n, 0);
    for (int e = 0; e < n; ++e)
      for (int i = n - 1; i >= 0; --i)
        {
          double s = Q[i * n + e];
          for (int k = i + 1; k < n; ++k)
            s -= L[k * n + i] * h.E[k * n + e];
          h.E[i * n + e] = s / L[i * n + i];
        }
    // Eigenvectors even ones first, odd ones behind (each group by ascending eigenvalue): the cell kernel
    // applies E in even-odd form (mul_E).  The operator is invariant under x -> 1 - x, so every
    // eigenvector of a simple eigenvalue has a parity; a pair that does not (degenerate eigenvalues) keeps
    // the ascending order and the dense product.
    {
      std::vector<int> parity(n, 0);
      bool             pure = true;
      for (int e = 0; e < n; ++e)
        {
          double even = 0, odd = 0, nrm = 0;
          for (int i = 0; i < n; ++i)
            {
              even += std::fabs(h.E[i * n + e] - h.E[(n - 1 - i) * n + e]);
              odd += std::fabs(h.E[i * n + e] + h.E[(n - 1 - i) * n + e]);
              nrm += std::fabs(h.E[i * n + e]);
            }
          parity[e] = even <= 1e-9 * nrm ? 1 : (odd <= 1e-9 * nrm ? -1 : 0);
          pure      = pure && parity[e] != 0;
        }
      const int n_even = (int)std::count(parity.begin(), parity.end(), 1);
      h.e_parity = pure && n_even == n - n / 2;
      if (h.e_parity)
        {
          std::vector<int> order;
          for (int pass = 1; pass >= -1; pass -= 2)
            for (int e = 0; e < n; ++e)
              if (parity[e] == pass)
                order.push_back(e);
          std::vector<double> E2(n * n), l2(n);
          for (int k = 0; k < n; ++k)
            {
              l2[k] = h.lambda[order[k]];
              for (int i = 0; i < n; ++i)
                E2[i * n + k] = h.E[i * n + order[k]];
            }
          h.E.swap(E2);
          h.lambda.swap(l2);
        }
    }
    // 1D forms in the eigenvector basis
    h.lt.assign(n, 0);
    h.ct.assign(n, 0);
    for (int s = 0; s < 2; ++s)
      {
        h.beta[s].assign(n, 0);
        h.gamma[s].assign(n, 0);
      }
    for (int e = 0; e < n; ++e)
      {
        for (int i = 0; i < n; ++i)
          for (int j = 0; j < n; ++j)
            {
              double l = 0;
              for (int q = 0; q < n; ++q)
                l += h.wq[q] * h.SD[q * n + i] * h.SD[q * n + j];
              h.lt[e] += h.E[i * n + e] * l * h.E[j * n + e];
              h.ct[e] += h.E[i * n + e] * cfirst[i * n + j] * h.E[j * n + e];
            }
        for (int s = 0; s < 2; ++s)
          for (int i = 0; i < n; ++i)
            {
              h.beta[s][e] += h.E[i * n + e] * h.fb[s][i];
              h.gamma[s][e] += h.E[i * n + e] * h.fg[s][i];
            }
      }
    return MGX_OK;
  }

  struct Geometry
  {
    double K[6], cn[3][3], fw[3], sigma[3];
  };

  int build_geometry(const double J[9], int p, Geometry &g, std::string &why)
  {
    const double det = J[0] * (J[4] * J[8] - J[5] * J[7]) - J[1] * (J[3] * J[8] - J[5] * J[6]) +
                       J[2] * (J[3] * J[7] - J[4] * J[6]);
    if (!(std::abs(det) > 0))
      {
        why = "singular cell Jacobian";
        return MGX_ERR_INVALID_ARGUMENT;
      }
    double inv[3][3]; // inv[a][i] = d xi_a / d x_i
    inv[0][0] = (J[4] * J[8] - J[5] * J[7]) / det;
    inv[0][1] = (J[2] * J[7] - J[1] * J[8]) / det;
    inv[0][2] = (J[1] * J[5] - J[2] * J[4]) / det;
    inv[1][0] = (J[5] * J[6] - J[3] * J[8]) / det;
    inv[1][1] = (J[0] * J[8] - J[2] * J[6]) / det;
    inv[1][2] = (J[2] * J[3] - J[0] * J[5]) / det;
    inv[2][0] = (J[3] * J[7] - J[4] * J[6]) / det;
    inv[2][1] = (J[1] * J[6] - J[0] * J[7]) / det;
    inv[2][2] = (J[0] * J[4] - J[1] * J[3]) / det;
    double G[3][3];
    for (int a = 0; a < 3; ++a)
      for (int c = 0; c < 3; ++c)
        G[a][c] = inv[a][0] * inv[c][0] + inv[a][1] * inv[c][1] + inv[a][2] * inv[c][2];
    const double ad = std::abs(det);
    g.K[0] = ad * G[0][0];
    g.K[1] = ad * G[1][1];
    g.K[2] = ad * G[2][2];
    g.K[3] = ad * G[0][1];
    g.K[4] = ad * G[0][2];
    g.K[5] = ad * G[1][2];
    for (int d = 0; d < 3; ++d)
      {
        const double nrm = std::sqrt(G[d][d]);
        for (int a = 0; a < 3; ++a)
          g.cn[d][a] = G[d][a] / nrm;
        g.fw[d]    = ad * nrm;
        g.sigma[d] = double(p + 1) * (p + 1) * std::abs(g.cn[d][d]); // penalty_factor = 1 (:47)
      }
    return MGX_OK;
  }

  // diagonal of T^T A_KK T for one combination of Dirichlet faces (bit f of cat): Kronecker
  // products of 1D forms in the eigenvector basis, in which the mass matrix is the identity
  void transformed_diagonal(const Host1D &h, const Geometry &g, unsigned cat, std::vector<double> &diag)
  {
    const int n = h.n;
    diag.assign((size_t)n * n * n, 0.0);
    for (int k = 0; k < n; ++k)
      for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i)
          {
            const int e[3] = {i, j, k};
            double    v    = g.K[0] * h.lt[i] + g.K[1] * h.lt[j] + g.K[2] * h.lt[k] +
                       2 * (g.K[3] * h.ct[i] * h.ct[j] + g.K[4] * h.ct[i] * h.ct[k] + g.K[5] * h.ct[j] * h.ct[k]);
            for (int f = 0; f < 6; ++f)
              {
                const int    d = f / 2, s = f % 2;
                const double fbnd = (cat >> f) & 1u ? 1.0 : 0.5;
                const double sgn  = s ? 1.0 : -1.0;
                const double be = h.beta[s][e[d]], ga = h.gamma[s][e[d]];
                double       vn = g.cn[d][d] * be * ga;
                for (int a = 0; a < 3; ++a)
                  if (a != d)
                    vn += g.cn[d][a] * be * be * h.ct[e[a]];
                v += g.fw[d] * (2 * fbnd * g.sigma[d] * be * be - 2 * fbnd * sgn * vn);
              }
            diag[(k * n + j) * n + i] = v;
          }
  }

  template <typename T>
  void fill_const(const Host1D &h, const Geometry &g, DGConst<T> &c)
  {
    std::memset(&c, 0, sizeof(c));
    const int n = h.n;
    for (int i = 0; i < n * n; ++i)
      {
        c.S[i] = (T)h.S[i];
        c.D[i] = (T)h.D[i];
        c.E[i] = (T)h.E[i];
      }
    for (int r = 0; r < n; ++r)
      for (int q = 0; q < n; ++q)
        {
          c.St[r * n + q] = (T)h.S[q * n + r];
          c.Dt[r * n + q] = (T)h.D[q * n + r];
          c.Et[r * n + q] = (T)h.E[q * n + r];
        }
    // even-odd tables (the symmetry itself is checked when the operator is created)
    auto eo_fill = [&](EOLine<T> &e, auto M) { // M(q, i)
      const int H = n / 2;
      for (int q = 0; q < H; ++q)
        for (int i = 0; i < H; ++i)
          {
            e.eo[2 * (q * H + i)]     = (T)(0.5 * (M(q, i) + M(n - 1 - q, i)));
            e.eo[2 * (q * H + i) + 1] = (T)(0.5 * (M(q, i) - M(n - 1 - q, i)));
          }
      if (n % 2)
        {
          for (int i = 0; i < H; ++i)
            {
              e.mrow[i] = (T)M(H, i);
              e.mcol[i] = (T)M(i, H);
            }
          e.mm = (T)M(H, H);
        }
    };
    {
      const int H = n / 2, Ne = n - H, No = H;
      c.eo_e      = h.e_parity ? 1 : 0;
      for (int i = 0; i < H; ++i)
        {
          for (int k = 0; k < No; ++k)
            {
              c.epair[2 * (i * No + k)]     = (T)h.E[i * n + k];
              c.epair[2 * (i * No + k) + 1] = (T)h.E[i * n + Ne + k];
            }
          c.elast[i] = (T)h.E[i * n + Ne - 1];
        }
      for (int e = 0; e < Ne; ++e)
        c.emid[e] = (n % 2) ? (T)h.E[H * n + e] : (T)0;
    }
    eo_fill(c.eoS, [&](int q, int i) { return h.S[q * n + i]; });
    eo_fill(c.eoSt, [&](int q, int i) { return h.S[i * n + q]; });
    eo_fill(c.eoD, [&](int q, int i) { return h.D[q * n + i]; });
    eo_fill(c.eoDt, [&](int q, int i) { return h.D[i * n + q]; });
    for (int i = 0; i < n; ++i)
      {
        c.w[i] = (T)h.wq[i];
        for (int s = 0; s < 2; ++s)
          {
            c.b[s][i]  = (T)h.b[s][i];
            c.g[s][i]  = (T)h.g[s][i];
            c.fb[s][i] = (T)h.fb[s][i];
            c.fg[s][i] = (T)h.fg[s][i];
          }
      }
    for (int i = 0; i < 6; ++i)
      c.K[i] = (T)g.K[i];
    for (int d = 0; d < 3; ++d)
      {
        for (int a = 0; a < 3; ++a)
          c.cn[d][a] = (T)g.cn[d][a];
        c.fw[d]    = (T)g.fw[d];
        c.sigma[d] = (T)g.sigma[d];
      }
    c.hderiv = (T)h.hderiv;
  }

  template <int P, typename T, int TYPE, int ACTION>
  void launch_one(hipStream_t s, const DGArgs<T> &a, bool ghosts)
  {
    using C             = DGCfg<P, T>;
    const uint32_t grid = (a.n_cells + C::CPW - 1) / C::CPW;
    if constexpr (TYPE == MGX_DG_HERMITE && ACTION != kJacobi)
      if (ghosts)
        {
          hipLaunchKernelGGL((dg_cell_kernel<P, T, TYPE, ACTION, true>), dim3(grid), dim3(C::THREADS), 0, s, a);
          return;
        }
    hipLaunchKernelGGL((dg_cell_kernel<P, T, TYPE, ACTION>), dim3(grid), dim3(C::THREADS), 0, s, a);
  }

  template <int P, typename T, int TYPE>
  void launch_action(hipStream_t s, int action, const DGArgs<T> &a, bool ghosts)
  {
    switch (action)
      {
        case kVmult:
          return launch_one<P, T, TYPE, kVmult>(s, a, ghosts);
        case kRestrict:
          return launch_one<P, T, TYPE, kRestrict>(s, a, ghosts);
        case kCgSums:
          return launch_one<P, T, TYPE, kCgSums>(s, a, ghosts);
        case kChebyshev:
          return launch_one<P, T, TYPE, kChebyshev>(s, a, ghosts);
        case kResidual:
          return launch_one<P, T, TYPE, kResidual>(s, a, ghosts);
        default:
          return launch_one<P, T, TYPE, kJacobi>(s, a, ghosts);
      }
  }

  template <int P, typename T>
  void launch_type(hipStream_t s, int basis, int action, const DGArgs<T> &a, bool ghosts)
  {
    if (basis == MGX_DG_HERMITE)
      launch_action<P, T, MGX_DG_HERMITE>(s, action, a, ghosts);
    else if (basis == MGX_DG_GAUSS_LOBATTO)
      launch_action<P, T, MGX_DG_GAUSS_LOBATTO>(s, action, a, ghosts);
    else
      launch_action<P, T, MGX_DG_GAUSS>(s, action, a, ghosts);
  }

  template <typename T>
  void launch_degree(hipStream_t s, int p, int basis, int action, const DGArgs<T> &a, bool ghosts)
  {
    switch (p)
      {
        case 1:
          return launch_type<1, T>(s, basis, action, a, ghosts);
        case 2:
          return launch_type<2, T>(s, basis, action, a, ghosts);
        case 3:
          return launch_type<3, T>(s, basis, action, a, ghosts);
        case 4:
          return launch_type<4, T>(s, basis, action, a, ghosts);
        case 5:
          return launch_type<5, T>(s, basis, action, a, ghosts);
        case 6:
          return launch_type<6, T>(s, basis, action, a, ghosts);
        case 7:
          return launch_type<7, T>(s, basis, action, a, ghosts);
        case 8:
          return launch_type<8, T>(s, basis, action, a, ghosts);
        default:
          return launch_type<9, T>(s, basis, action, a, ghosts);
      }
  }
} // namespace

struct mgx_dg_operator_s
{
  mgx_context_t ctx    = nullptr;
  int           degree = 0, basis = 0, number = MGX_F32;
  uint32_t      n_cells = 0;
  int32_t      *neigh   = nullptr; // device
  void         *consts  = nullptr; // device DGConst<T>
  void         *inv_diag = nullptr; // device [64][(p+1)^3]
  Host1D        h;
  Geometry      g;
  // decomposed mesh: ghost cells behind the owned ones, filled from their owners before every
  // application (mgx_dg_update_ghost_values)
  uint32_t                n_ghost = 0;
  int                     plan_id = 0;
  std::vector<int>        nb_rank;
  std::vector<uint32_t>   nb_count, nb_recv_first, nb_entries;
  std::vector<uint32_t *> nb_cells_dev;
  std::vector<void *>     nb_send;
  // cells without / with a ghost neighbour: the former run while the ghost exchange is in flight
  // (the reference completes its exchange, laplace_operator_dg.h:986-1057, before the cell loop)
  uint32_t *interior_cells = nullptr, *boundary_cells = nullptr; // device
  uint32_t  n_interior = 0, n_boundary = 0;
  bool      interior_is_prefix = false; // the interior cells are cells 0 ... n_interior - 1
  // entries per ghost: a whole cell, or for the Hermite-like basis two values per face point
  // (data_per_face of laplace_operator_dg.h:565)
  uint32_t               ghost_stride = 0;
  std::vector<uint8_t *> nb_faces_dev; // Hermite-like basis: face of every sent cell towards the neighbour rank
  // block sums of the merged CG iteration (action 2) and their total, allocated at the first use
  double  *cg_partials = nullptr, *cg_sums = nullptr;
  uint32_t cg_capacity = 0;
};

// MultigridSolverDG (common/multigrid_solver_dg.h:55-747): the DG level on top of an FE_Q hierarchy
struct mgx_dg_solver_s
{
  mgx_context_t     ctx = nullptr;
  mgx_dg_operator_t A = nullptr, A_dp = nullptr;
  mgx_solver_t      cfe = nullptr;
  int               degree = 0, number = MGX_F32;
  size_t            n = 0;      // owned DoFs
  size_t            n_vec = 0;  // entries of a vector: owned cells, then ghost cells
  mgx_operator_t    fe = nullptr; // finest FE_Q operator (interface sum of the restricted defect)
  bool              decomposed = false;
  mgx_smoother_info info{};
  void             *defect = nullptr, *t = nullptr, *update = nullptr, *old = nullptr; // V-cycle number type
  void             *P1 = nullptr;                                                      // device, V-cycle number type
  const uint32_t   *idx27 = nullptr;
  uint32_t          n_cells = 0, n_cg = 0;
  bool              cg_eight_colours = false; // cells c, c + 8, ... of the FE_Q level share no DoF
  void             *cg_defect = nullptr, *cg_update = nullptr; // the FE_Q solver's finest-level vectors
  double           *r = nullptr, *z = nullptr, *d = nullptr, *h = nullptr; // PCG, fp64
};

namespace
{
  int dg_fail(int code, const std::string &msg) { return mgx::report_error(code, msg.c_str()); }

#define DG_HIP(call)                                                                       \
  do                                                                                       \
    {                                                                                      \
      hipError_t e_ = (call);                                                              \
      if (e_ != hipSuccess)                                                                \
        return dg_fail(MGX_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));    \
    }                                                                                      \
  while (0)

#define MGX_DG_TRY(call) \
  do                     \
    {                    \
      int s_ = (call);   \
      if (s_ != MGX_OK)  \
        return s_;       \
    }                    \
  while (0)

  template <typename T>
  __global__ void __launch_bounds__(256)
    k_pack_cells(T *__restrict__ buf, const T *__restrict__ vec, const uint32_t *__restrict__ cells, uint32_t count,
                 uint32_t n3)
  {
    const uint64_t total = (uint64_t)count * n3;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x)
      {
        const uint32_t c = (uint32_t)(i / n3), k = (uint32_t)(i - (uint64_t)c * n3);
        buf[i] = vec[(uint64_t)cells[c] * n3 + k];
      }
  }

  // Hermite-like basis: what the neighbour needs of a cell is the value and the normal derivative
  // on the shared face, from the two node layers next to it (laplace_operator_dg.h:1015-1039)
  template <typename T>
  __global__ void __launch_bounds__(256)
    k_pack_faces(T *__restrict__ buf, const T *__restrict__ vec, const uint32_t *__restrict__ cells,
                 const uint8_t *__restrict__ faces, uint32_t count, int N, T hderiv)
  {
    const uint32_t nn2   = (uint32_t)(N * N);
    const uint64_t total = (uint64_t)count * nn2;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x)
      {
        const uint32_t c = (uint32_t)(i / nn2), j = (uint32_t)(i - (uint64_t)c * nn2);
        const int      a = (int)(j % N), b = (int)(j / N), f = faces[c], d = f / 2, upper = f % 2;
        const int      sd = d == 0 ? 1 : (d == 1 ? N : N * N), s1 = d == 0 ? N : 1, s2 = d == 2 ? N : N * N;
        const T *__restrict__ x = vec + (uint64_t)cells[c] * nn2 * N + a * s1 + b * s2;
        const T v0 = x[(upper ? N - 1 : 0) * sd];
        const T v1 = x[(upper ? (N > 1 ? N - 2 : 0) : (N > 1 ? 1 : 0)) * sd];
        buf[2 * i]     = v0;
        buf[2 * i + 1] = upper ? hderiv * (v1 - v0) : hderiv * (v0 - v1);
      }
  }

  // pack kernels on the context's stream; `overlap`: the exchange itself on the side stream, begun
  // behind the pack kernels -- the caller enqueues independent work and then calls ghosts_finish
  int ghosts_pack(mgx_dg_operator_t op, const void *vec)
  {
    hipStream_t    s   = (hipStream_t)mgx_context_stream(op->ctx);
    const uint32_t n3  = (uint32_t)(op->degree + 1) * (op->degree + 1) * (op->degree + 1);
    const int      nnb = (int)op->nb_rank.size();
    for (int k = 0; k < nnb; ++k)
      {
        const bool     faces = op->basis == MGX_DG_HERMITE;
        const int      N     = op->degree + 1;
        const uint64_t total = (uint64_t)op->nb_count[k] * (faces ? (uint32_t)(N * N) : n3);
        const uint32_t grid  = (uint32_t)std::min<uint64_t>((total + 255) / 256, 4096);
        if (faces && op->number == MGX_F64)
          hipLaunchKernelGGL(k_pack_faces<double>, dim3(grid), dim3(256), 0, s, (double *)op->nb_send[k], (const double *)vec,
                             op->nb_cells_dev[k], op->nb_faces_dev[k], op->nb_count[k], N, op->h.hderiv);
        else if (faces)
          hipLaunchKernelGGL(k_pack_faces<float>, dim3(grid), dim3(256), 0, s, (float *)op->nb_send[k], (const float *)vec,
                             op->nb_cells_dev[k], op->nb_faces_dev[k], op->nb_count[k], N, (float)op->h.hderiv);
        else if (op->number == MGX_F64)
          hipLaunchKernelGGL(k_pack_cells<double>, dim3(grid), dim3(256), 0, s, (double *)op->nb_send[k], (const double *)vec,
                             op->nb_cells_dev[k], op->nb_count[k], n3);
        else
          hipLaunchKernelGGL(k_pack_cells<float>, dim3(grid), dim3(256), 0, s, (float *)op->nb_send[k], (const float *)vec,
                             op->nb_cells_dev[k], op->nb_count[k], n3);
      }
    DG_HIP(hipGetLastError());
    return MGX_OK;
  }

  int ghosts_exchange(mgx_dg_operator_t op, void *vec, hipStream_t stream)
  {
    const uint32_t n3  = (uint32_t)(op->degree + 1) * (op->degree + 1) * (op->degree + 1);
    const size_t   es  = op->number == MGX_F64 ? 8 : 4;
    const int      nnb = (int)op->nb_rank.size();
    std::vector<void *> recv(nnb);
    for (int k = 0; k < nnb; ++k)
      recv[k] = (char *)vec + ((size_t)op->n_cells * n3 + (size_t)(op->nb_recv_first[k] - op->n_cells) * op->ghost_stride) *
                                es; // straight into the ghosts
    return mgx::exchange_buffers(op->ctx, op->plan_id, op->number, nnb, op->nb_rank.data(), op->nb_entries.data(),
                                 op->nb_send.data(), recv.data(), stream);
  }

  int update_ghosts(mgx_dg_operator_t op, void *vec)
  {
    if (op->n_ghost == 0)
      return MGX_OK;
    MGX_DG_TRY(ghosts_pack(op, vec));
    return ghosts_exchange(op, vec, nullptr);
  }

  // operands of the merged actions 1 and 2 (DGArgs)
  struct MergedArgs
  {
    uint32_t        cell_stride = 1;
    double         *partials    = nullptr;
    void           *cg          = nullptr;
    const uint32_t *idx27       = nullptr;
    const void     *P1          = nullptr;
    int             plain       = 0;
  };

  template <typename T>
  uint32_t dg_grid(int p, uint32_t n_cells)
  {
    static const uint32_t cpw[10] = {1, DGCfg<1, T>::CPW, DGCfg<2, T>::CPW, DGCfg<3, T>::CPW, DGCfg<4, T>::CPW, DGCfg<5, T>::CPW,
                                     DGCfg<6, T>::CPW, DGCfg<7, T>::CPW, DGCfg<8, T>::CPW, DGCfg<9, T>::CPW};
    return (n_cells + cpw[p] - 1) / cpw[p];
  }

  int launch_cells(mgx_dg_operator_t op, int action, void *dst, const void *rhs, const void *src, double f1, double f2,
                   int iteration_index, const uint32_t *cell_list, uint32_t n_cells, hipStream_t stream = nullptr,
                   uint32_t cell_first = 0, const MergedArgs &m = MergedArgs())
  {
    if (n_cells == 0)
      return MGX_OK;
    hipStream_t s = stream ? stream : (hipStream_t)mgx_context_stream(op->ctx);
    if (op->number == MGX_F64)
      {
        DGArgs<double> a{(const double *)src, (const double *)rhs, (double *)dst, op->neigh,
                         (const DGConst<double> *)op->consts, (const double *)op->inv_diag, cell_list, cell_first, n_cells, op->n_cells, f1, f2,
                         iteration_index, m.cell_stride, m.partials, (double *)m.cg, m.idx27, (const double *)m.P1, m.plain};
        launch_degree<double>(s, op->degree, op->basis, action, a, op->n_ghost > 0);
      }
    else
      {
        DGArgs<float> a{(const float *)src, (const float *)rhs, (float *)dst, op->neigh,
                        (const DGConst<float> *)op->consts, (const float *)op->inv_diag, cell_list, cell_first, n_cells, op->n_cells, (float)f1,
                        (float)f2, iteration_index, m.cell_stride, m.partials, (float *)m.cg, m.idx27, (const float *)m.P1, m.plain};
        launch_degree<float>(s, op->degree, op->basis, action, a, op->n_ghost > 0);
      }
    DG_HIP(hipGetLastError());
    return MGX_OK;
  }

  // One application.  with_ghosts: the action reads neighbour cells, so the ghost cells of src are
  // refreshed first; the cells without a ghost neighbour run while that exchange is in flight on the
  // context's side stream (with the blocking callback transport: while the host waits in it).
  int run(mgx_dg_operator_t op, int action, void *dst, const void *rhs, const void *src, double f1, double f2,
          int iteration_index, bool with_ghosts = false, const MergedArgs &m = MergedArgs())
  {
    if (!with_ghosts || op->n_ghost == 0)
      return launch_cells(op, action, dst, rhs, src, f1, f2, iteration_index, nullptr, op->n_cells, nullptr, 0, m);
    void *ghosted = const_cast<void *>(src);
    MGX_DG_TRY(ghosts_pack(op, src));
    hipStream_t side = (op->n_interior > 0 && !mgx::context_tunables(op->ctx).dg_no_overlap) ? mgx::side_stream_begin(op->ctx)
                                                                                               : nullptr;
    if (!side)
      {
        MGX_DG_TRY(ghosts_exchange(op, ghosted, nullptr));
        return launch_cells(op, action, dst, rhs, src, f1, f2, iteration_index, nullptr, op->n_cells, nullptr, 0, m);
      }
    // main stream: interior cells; side stream: exchange, then the cells next to a ghost cell (they
    // write other cells of dst than the interior launch and share its read-only operands)
    // (interior cells first in the caller's order: two contiguous ranges, no index lists)
    const uint32_t *li = op->interior_is_prefix ? nullptr : op->interior_cells;
    const uint32_t *lb = op->interior_is_prefix ? nullptr : op->boundary_cells;
    // whatever fails below, the main stream is ordered behind the side stream again before returning:
    // nothing of this application may still be in flight when the caller reuses src / dst
    int status = launch_cells(op, action, dst, rhs, src, f1, f2, iteration_index, li, op->n_interior, nullptr, 0, m);
    if (status == MGX_OK)
      status = ghosts_exchange(op, ghosted, side);
    MergedArgs mb = m; // the block sums of the second launch behind those of the first
    if (m.partials)
      mb.partials += 4 * (size_t)(op->number == MGX_F64 ? dg_grid<double>(op->degree, op->n_interior)
                                                        : dg_grid<float>(op->degree, op->n_interior));
    if (status == MGX_OK)
      status = launch_cells(op, action, dst, rhs, src, f1, f2, iteration_index, lb, op->n_boundary, side,
                            op->interior_is_prefix ? op->n_interior : 0, mb);
    const int joined = mgx::side_stream_end(op->ctx);
    return status != MGX_OK ? status : joined;
  }
} // namespace

extern "C" {

int mgx_dg_operator_create(mgx_context_t ctx, const mgx_dg_operator_desc *desc, mgx_dg_operator_t *out)
{
  if (!ctx || !desc || !out)
    return dg_fail(MGX_ERR_INVALID_ARGUMENT, "mgx_dg_operator_create: null argument");
  if (desc->degree < 1 || desc->degree > MGX_MAX_DEGREE)
    return dg_fail(MGX_ERR_UNSUPPORTED, "mgx_dg_operator_create: degree must be in 1.." + std::to_string(MGX_MAX_DEGREE));
  if (desc->basis < MGX_DG_HERMITE || desc->basis > MGX_DG_GAUSS)
    return dg_fail(MGX_ERR_UNSUPPORTED, "mgx_dg_operator_create: basis must be MGX_DG_HERMITE, _GAUSS_LOBATTO or _GAUSS");
  if (desc->number != MGX_F32 && desc->number != MGX_F64)
    return dg_fail(MGX_ERR_INVALID_ARGUMENT, "mgx_dg_operator_create: number must be MGX_F32 or MGX_F64");
  if (desc->n_cells == 0 || !desc->neighbours)
    return dg_fail(MGX_ERR_INVALID_ARGUMENT, "mgx_dg_operator_create: empty mesh");
  const uint64_t n3 = (uint64_t)(desc->degree + 1) * (desc->degree + 1) * (desc->degree + 1);
  if ((uint64_t)desc->n_cells * n3 * (desc->number == MGX_F64 ? 8 : 4) >= (1ull << 40))
    return dg_fail(MGX_ERR_UNSUPPORTED, "mgx_dg_operator_create: vector larger than 1 TiB");
  const uint64_t n_all = (uint64_t)desc->n_cells + desc->n_ghost_cells;
  if (n_all * n3 >= (1ull << 32))
    return dg_fail(MGX_ERR_UNSUPPORTED, "mgx_dg_operator_create: more than 2^32 vector entries per rank");
  for (uint64_t i = 0; i < (uint64_t)desc->n_cells * 6; ++i)
    if (desc->neighbours[i] != MGX_DG_BOUNDARY && (desc->neighbours[i] < 0 || (uint64_t)desc->neighbours[i] >= n_all))
      return dg_fail(MGX_ERR_INVALID_ARGUMENT, "mgx_dg_operator_create: neighbour entry " + std::to_string(i) +
                                                 " is neither a cell of the mesh, a ghost cell nor MGX_DG_BOUNDARY");
  if (desc->n_ghost_cells > 0)
    {
      if (!desc->exchange || !mgx::context_has_comm(ctx))
        return dg_fail(MGX_ERR_INVALID_ARGUMENT, "mgx_dg_operator_create: ghost cells need an exchange plan and a "
                                                 "communicator on the context");
      const mgx_dg_exchange_desc &e = *desc->exchange;
      if (e.n_neighbors < 1 || !e.neighbor_rank || !e.count || !e.send_cells || !e.recv_first)
        return dg_fail(MGX_ERR_INVALID_ARGUMENT, "mgx_dg_operator_create: incomplete exchange plan");
      uint64_t                    covered = 0;
      for (int k = 0; k < e.n_neighbors; ++k)
        {
          if (k > 0 && e.neighbor_rank[k] <= e.neighbor_rank[k - 1])
            return dg_fail(MGX_ERR_INVALID_ARGUMENT, "mgx_dg_operator_create: neighbour ranks must be ascending");
          if (e.recv_first[k] < desc->n_cells || (uint64_t)e.recv_first[k] + e.count[k] > n_all)
            return dg_fail(MGX_ERR_INVALID_ARGUMENT, "mgx_dg_operator_create: ghost range outside the ghost cells");
          if (e.count[k] > 0 && !e.send_cells[k])
            return dg_fail(MGX_ERR_INVALID_ARGUMENT, "mgx_dg_operator_create: incomplete exchange plan");
          for (uint32_t i = 0; i < e.count[k]; ++i)
            if (e.send_cells[k][i] >= desc->n_cells)
              return dg_fail(MGX_ERR_INVALID_ARGUMENT, "mgx_dg_operator_create: only owned cells can be sent");
          covered += e.count[k];
        }
      if (covered != desc->n_ghost_cells)
        return dg_fail(MGX_ERR_INVALID_ARGUMENT, "mgx_dg_operator_create: the exchange plan does not fill every ghost cell "
                                                 "exactly once");
    }
  std::unique_ptr<mgx_dg_operator_s> op(new mgx_dg_operator_s);
  op->ctx     = ctx;
  op->degree  = desc->degree;
  op->basis   = desc->basis;
  op->number  = desc->number;
  op->n_cells = desc->n_cells;
  op->n_ghost = desc->n_ghost_cells;
  op->ghost_stride = desc->basis == MGX_DG_HERMITE ? 2u * (desc->degree + 1) * (desc->degree + 1) : (uint32_t)n3;
  std::string why;
  int         status = build_1d(desc->degree, desc->basis, op->h, why);
  if (status == MGX_OK)
    status = build_geometry(desc->jacobian, desc->degree, op->g, why);
  if (status != MGX_OK)
    return dg_fail(status, "mgx_dg_operator_create: " + why);
  {
    // the even-odd line products of the cell kernel (mul_eo) rest on the reversal symmetry of the 1D
    // matrices: S[q][i] = S[n-1-q][n-1-i], D[q][r] = -D[n-1-q][n-1-r].  All three bases have it.
    const int n = op->h.n;
    double    dev = 0, scale = 0;
    for (int q = 0; q < n; ++q)
      for (int i = 0; i < n; ++i)
        {
          dev   = std::max(dev, std::fabs(op->h.S[q * n + i] - op->h.S[(n - 1 - q) * n + n - 1 - i]));
          dev   = std::max(dev, std::fabs(op->h.D[q * n + i] + op->h.D[(n - 1 - q) * n + n - 1 - i]));
          scale = std::max(scale, std::max(std::fabs(op->h.S[q * n + i]), std::fabs(op->h.D[q * n + i])));
        }
    if (dev > 1e-11 * scale)
      return dg_fail(MGX_ERR_UNSUPPORTED, "mgx_dg_operator_create: the 1D basis is not symmetric under x -> 1 - x");
    if (!op->h.e_parity)
      return dg_fail(MGX_ERR_UNSUPPORTED, "mgx_dg_operator_create: the eigenvectors of the 1D problem have no definite parity "
                                          "(degenerate eigenvalues)");
  }

  const size_t        nsz = desc->number == MGX_F64 ? 8 : 4;
  std::vector<double> table(64 * n3), diag;
  for (unsigned cat = 0; cat < 64; ++cat)
    {
      transformed_diagonal(op->h, op->g, cat, diag);
      for (uint64_t i = 0; i < n3; ++i)
        {
          if (!(diag[i] > 0))
            return dg_fail(MGX_ERR_UNSUPPORTED, "mgx_dg_operator_create: transformed cell block is not positive");
          table[cat * n3 + i] = 1.0 / diag[i];
        }
    }
  hipStream_t s = (hipStream_t)mgx_context_stream(ctx);
  auto        cleanup = [&]() {
    (void)hipFree(op->neigh);
    (void)hipFree(op->consts);
    (void)hipFree(op->inv_diag);
    (void)hipFree(op->interior_cells);
    (void)hipFree(op->boundary_cells);
    for (auto *p : op->nb_cells_dev)
      (void)hipFree(p);
    for (auto *p : op->nb_send)
      (void)hipFree(p);
    for (auto *p : op->nb_faces_dev)
      (void)hipFree(p);
  };
#define DG_HIP_C(call)                                                                      \
  do                                                                                        \
    {                                                                                       \
      hipError_t e_ = (call);                                                               \
      if (e_ != hipSuccess)                                                                 \
        {                                                                                   \
          cleanup();                                                                        \
          return dg_fail(MGX_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));   \
        }                                                                                   \
    }                                                                                       \
  while (0)
  DG_HIP_C(hipMalloc((void **)&op->neigh, sizeof(int32_t) * 6 * (size_t)desc->n_cells));
  DG_HIP_C(hipMemcpyAsync(op->neigh, desc->neighbours, sizeof(int32_t) * 6 * (size_t)desc->n_cells,
                          hipMemcpyHostToDevice, s));
  if (op->n_ghost > 0)
    {
      const mgx_dg_exchange_desc &e = *desc->exchange;
      op->plan_id                   = e.plan_id;
      for (int k = 0; k < e.n_neighbors; ++k)
        {
          op->nb_rank.push_back(e.neighbor_rank[k]);
          op->nb_count.push_back(e.count[k]);
          op->nb_entries.push_back((uint32_t)((uint64_t)e.count[k] * op->ghost_stride));
          op->nb_recv_first.push_back(e.recv_first[k]);
          uint32_t *cells = nullptr;
          void     *buf   = nullptr;
          DG_HIP_C(hipMalloc((void **)&cells, sizeof(uint32_t) * ((size_t)e.count[k] + 1)));
          op->nb_cells_dev.push_back(cells);
          DG_HIP_C(hipMemcpyAsync(cells, e.send_cells[k], sizeof(uint32_t) * e.count[k], hipMemcpyHostToDevice, s));
          DG_HIP_C(hipMalloc(&buf, nsz * ((size_t)e.count[k] * op->ghost_stride + 1)));
          op->nb_send.push_back(buf);
          if (op->basis == MGX_DG_HERMITE)
            {
              // the face of every sent cell that looks at this neighbour's cells
              std::vector<uint8_t> face(e.count[k]);
              for (uint32_t i = 0; i < e.count[k]; ++i)
                {
                  int found = -1, n_found = 0;
                  for (int f = 0; f < 6; ++f)
                    {
                      const int32_t nbr = desc->neighbours[(size_t)e.send_cells[k][i] * 6 + f];
                      if (nbr >= 0 && (uint32_t)nbr >= e.recv_first[k] && (uint32_t)nbr < e.recv_first[k] + e.count[k])
                        {
                          found = f;
                          ++n_found;
                        }
                    }
                  if (n_found != 1)
                    {
                      cleanup();
                      return dg_fail(MGX_ERR_UNSUPPORTED, "mgx_dg_operator_create: a sent cell must touch the cells of the "
                                                          "receiving rank through exactly one face");
                    }
                  face[i] = (uint8_t)found;
                }
              uint8_t *fd = nullptr;
              DG_HIP_C(hipMalloc((void **)&fd, face.size() + 1));
              op->nb_faces_dev.push_back(fd);
              DG_HIP_C(hipMemcpy(fd, face.data(), face.size(), hipMemcpyHostToDevice));
            }
        }
    }
  if (op->n_ghost > 0)
    {
      std::vector<uint32_t> interior, boundary;
      for (uint32_t c = 0; c < desc->n_cells; ++c)
        {
          bool ghost = false;
          for (int f = 0; f < 6; ++f)
            ghost = ghost || (desc->neighbours[(size_t)c * 6 + f] >= 0 && (uint32_t)desc->neighbours[(size_t)c * 6 + f] >= desc->n_cells);
          (ghost ? boundary : interior).push_back(c);
        }
      op->n_interior = (uint32_t)interior.size();
      op->n_boundary = (uint32_t)boundary.size();
      op->interior_is_prefix = interior.empty() || interior.back() + 1 == interior.size();
      DG_HIP_C(hipMalloc((void **)&op->interior_cells, sizeof(uint32_t) * (interior.size() + 1)));
      DG_HIP_C(hipMalloc((void **)&op->boundary_cells, sizeof(uint32_t) * (boundary.size() + 1)));
      DG_HIP_C(hipMemcpy(op->interior_cells, interior.data(), sizeof(uint32_t) * interior.size(), hipMemcpyHostToDevice));
      DG_HIP_C(hipMemcpy(op->boundary_cells, boundary.data(), sizeof(uint32_t) * boundary.size(), hipMemcpyHostToDevice));
    }
  DG_HIP_C(hipMalloc(&op->inv_diag, nsz * table.size()));
  if (desc->number == MGX_F64)
    {
      DGConst<double> c;
      fill_const(op->h, op->g, c);
      DG_HIP_C(hipMalloc(&op->consts, sizeof(c)));
      DG_HIP_C(hipMemcpyAsync(op->consts, &c, sizeof(c), hipMemcpyHostToDevice, s));
      DG_HIP_C(hipMemcpyAsync(op->inv_diag, table.data(), nsz * table.size(), hipMemcpyHostToDevice, s));
      DG_HIP_C(hipStreamSynchronize(s));
    }
  else
    {
      DGConst<float> c;
      fill_const(op->h, op->g, c);
      std::vector<float> tf(table.begin(), table.end());
      DG_HIP_C(hipMalloc(&op->consts, sizeof(c)));
      DG_HIP_C(hipMemcpyAsync(op->consts, &c, sizeof(c), hipMemcpyHostToDevice, s));
      DG_HIP_C(hipMemcpyAsync(op->inv_diag, tf.data(), nsz * tf.size(), hipMemcpyHostToDevice, s));
      DG_HIP_C(hipStreamSynchronize(s));
    }
#undef DG_HIP_C
  *out = op.release();
  return MGX_OK;
}

int mgx_dg_operator_destroy(mgx_dg_operator_t op)
{
  if (!op)
    return MGX_OK;
  (void)hipStreamSynchronize((hipStream_t)mgx_context_stream(op->ctx));
  (void)hipFree(op->neigh);
  (void)hipFree(op->consts);
  (void)hipFree(op->inv_diag);
  (void)hipFree(op->interior_cells);
  (void)hipFree(op->boundary_cells);
  (void)hipFree(op->cg_partials);
  (void)hipFree(op->cg_sums);
  for (auto *p : op->nb_cells_dev)
    (void)hipFree(p);
  for (auto *p : op->nb_send)
    (void)hipFree(p);
  for (auto *p : op->nb_faces_dev)
    (void)hipFree(p);
  delete op;
  return MGX_OK;
}

uint64_t mgx_dg_operator_vector_size(mgx_dg_operator_t op)
{
  return op ? (uint64_t)op->n_cells * (op->degree + 1) * (op->degree + 1) * (op->degree + 1) + (uint64_t)op->n_ghost * op->ghost_stride
            : 0;
}

int mgx_dg_update_ghost_values(mgx_dg_operator_t op, void *vec)
{
  if (!op || !vec)
    return dg_fail(MGX_ERR_INVALID_ARGUMENT, "mgx_dg_update_ghost_values: null argument");
  return update_ghosts(op, vec);
}

uint64_t mgx_dg_operator_n_dofs(mgx_dg_operator_t op)
{
  return op ? (uint64_t)op->n_cells * (op->degree + 1) * (op->degree + 1) * (op->degree + 1) : 0;
}

int mgx_dg_vmult(mgx_dg_operator_t op, void *dst, const void *src)
{
  if (!op || !dst || !src || dst == src)
    return dg_fail(MGX_ERR_INVALID_ARGUMENT, "mgx_dg_vmult: null or aliased vectors");
  return run(op, kVmult, dst, nullptr, src, 0, 0, 0, true);
}

int mgx_dg_vmult_residual(mgx_dg_operator_t op, void *dst, const void *rhs, const void *src)
{
  if (!op || !dst || !src || !rhs || dst == src)
    return dg_fail(MGX_ERR_INVALID_ARGUMENT, "mgx_dg_vmult_residual: null or aliased vectors");
  return run(op, kResidual, dst, rhs, src, 0, 0, 0, true);
}

int mgx_dg_jacobi_vmult(mgx_dg_operator_t op, void *dst, const void *src)
{
  if (!op || !dst || !src)
    return dg_fail(MGX_ERR_INVALID_ARGUMENT, "mgx_dg_jacobi_vmult: null vector");
  return run(op, kJacobi, dst, nullptr, src, 0, 1.0, 0);
}

int mgx_dg_vmult_with_chebyshev_update(mgx_dg_operator_t op, const void *rhs, unsigned iteration_index, double factor1,
                                       double factor2, void *solution, void *solution_old)
{
  if (!op || !rhs || !solution)
    return dg_fail(MGX_ERR_INVALID_ARGUMENT, "mgx_dg_vmult_with_chebyshev_update: null vector");
  if (iteration_index == 0)
    return run(op, kJacobi, solution, nullptr, rhs, 0, factor2, 0);
  if (!solution_old || solution_old == solution)
    return dg_fail(MGX_ERR_INVALID_ARGUMENT, "mgx_dg_vmult_with_chebyshev_update: solution_old is null or aliases solution");
  return run(op, kChebyshev, solution_old, rhs, solution, factor1, factor2, (int)iteration_index, true);
}

int mgx_dg_vmult_with_cg_update(mgx_dg_operator_t op, double alpha, double beta, const void *r, void *q, void *p, void *x,
                                double sums[4])
{
  if (!op || !r || !q || !p || !x || !sums || q == p)
    return dg_fail(MGX_ERR_INVALID_ARGUMENT, "mgx_dg_vmult_with_cg_update: null or aliased vectors");
  hipStream_t    s      = (hipStream_t)mgx_context_stream(op->ctx);
  const bool     f64    = op->number == MGX_F64;
  const bool     split  = op->n_ghost > 0 && op->n_interior > 0;
  const uint32_t blocks = f64 ? (split ? dg_grid<double>(op->degree, op->n_interior) + dg_grid<double>(op->degree, op->n_boundary)
                                       : dg_grid<double>(op->degree, op->n_cells))
                              : (split ? dg_grid<float>(op->degree, op->n_interior) + dg_grid<float>(op->degree, op->n_boundary)
                                       : dg_grid<float>(op->degree, op->n_cells));
  if (blocks > op->cg_capacity)
    {
      if (op->cg_partials)
        DG_HIP(hipFree(op->cg_partials));
      op->cg_partials = nullptr;
      op->cg_capacity = 0;
      DG_HIP(hipMalloc(&op->cg_partials, sizeof(double) * 4 * (size_t)blocks));
      op->cg_capacity = blocks;
    }
  if (!op->cg_sums)
    DG_HIP(hipMalloc(&op->cg_sums, sizeof(double) * 4));
  // laplace_operator_dg.h:871-902: x += alpha p ; p = beta p + q (alpha == 0: p = q) on the owned entries
  mgx::launch_cg_pre(s, op->number, x, p, q, alpha, beta, (size_t)mgx_dg_operator_n_dofs(op));
  // :903 q = A p with the sums of the next iteration
  if (op->n_cells == 0)
    DG_HIP(hipMemsetAsync(op->cg_sums, 0, sizeof(double) * 4, s));
  else
    {
      // without the overlap of the ghost exchange the cells run in one launch: its blocks are not the split's
      DG_HIP(hipMemsetAsync(op->cg_partials, 0, sizeof(double) * 4 * (size_t)blocks, s));
      MergedArgs m;
      m.partials = op->cg_partials;
      MGX_DG_TRY(run(op, kCgSums, q, r, p, 0, 0, 0, true, m));
      mgx::launch_reduce4(s, op->cg_partials, blocks, nullptr, op->cg_sums);
    }
  DG_HIP(hipMemcpyAsync(sums, op->cg_sums, sizeof(double) * 4, hipMemcpyDeviceToHost, s));
  DG_HIP(hipStreamSynchronize(s));
  return mgx::allreduce_sum(op->ctx, sums, 4); // :904-906
}

int mgx_dg_operator_info(mgx_dg_operator_t op, double *hderiv, double penalty[3], double eigenvalues_1d[MGX_MAX_DEGREE + 1])
{
  if (!op)
    return dg_fail(MGX_ERR_INVALID_ARGUMENT, "mgx_dg_operator_info: null operator");
  if (hderiv)
    *hderiv = op->h.hderiv;
  if (penalty)
    for (int d = 0; d < 3; ++d)
      penalty[d] = op->g.sigma[d];
  if (eigenvalues_1d)
    {
      std::vector<double> sorted(op->h.lambda.begin(), op->h.lambda.begin() + op->h.n);
      std::sort(sorted.begin(), sorted.end()); // (held parity by parity internally)
      for (int i = 0; i <= MGX_MAX_DEGREE; ++i)
        eigenvalues_1d[i] = i < op->h.n ? sorted[i] : 0.0;
    }
  return MGX_OK;
}

int mgx_dg_operator_basis(mgx_dg_operator_t op, double *shape_values, double *quadrature_points,
                          double *quadrature_weights)
{
  if (!op)
    return dg_fail(MGX_ERR_INVALID_ARGUMENT, "mgx_dg_operator_basis: null operator");
  const int n = op->h.n;
  if (shape_values)
    std::copy(op->h.S.begin(), op->h.S.begin() + n * n, shape_values);
  if (quadrature_points)
    std::copy(op->h.xq.begin(), op->h.xq.begin() + n, quadrature_points);
  if (quadrature_weights)
    std::copy(op->h.wq.begin(), op->h.wq.begin() + n, quadrature_weights);
  return MGX_OK;
}

int mgx_dg_cheby_mesh(int n_cell_steps, int cells[3], double jacobian[9])
{
  if (n_cell_steps < 0 || n_cell_steps > 30 || !cells || !jacobian)
    return dg_fail(MGX_ERR_INVALID_ARGUMENT, "mgx_dg_cheby_mesh: invalid argument");
  for (int d = 0; d < 3; ++d)
    {
      const double left = -1.0 + 0.05 * (d + 1), right = 0.95 - 0.06 * d;
      cells[d]          = (d < n_cell_steps % 3 ? 2 : 1) << (n_cell_steps / 3);
      const double h    = (right - left) / cells[d];
      for (int r = 0; r < 3; ++r)
        jacobian[r * 3 + d] = ((r == d ? 1.0 : 0.0) + 0.12 * (r + 1) * (d + 1)) * h;
    }
  return MGX_OK;
}

int mgx_dg_box_neighbours(const int cells[3], int ordering, int32_t *neighbours, int32_t *cell_ijk)
{
  if (!cells || !neighbours || cells[0] < 1 || cells[1] < 1 || cells[2] < 1)
    return dg_fail(MGX_ERR_INVALID_ARGUMENT, "mgx_dg_box_neighbours: invalid argument");
  const uint64_t n = (uint64_t)cells[0] * cells[1] * cells[2];
  if (n >= (1ull << 31))
    return dg_fail(MGX_ERR_UNSUPPORTED, "mgx_dg_box_neighbours: more than 2^31 cells");
  std::vector<uint32_t> order(n), position(n);
  std::iota(order.begin(), order.end(), 0u);
  auto ijk = [&](uint32_t lex, int out[3]) {
    out[0] = lex % cells[0];
    out[1] = (lex / cells[0]) % cells[1];
    out[2] = lex / ((uint64_t)cells[0] * cells[1]);
  };
  if (ordering == 1)
    {
      auto spread = [](uint64_t v) { // bits of v to every third position
        uint64_t r = 0;
        for (int bit = 0; bit < 21; ++bit)
          r |= ((v >> bit) & 1ull) << (3 * bit);
        return r;
      };
      std::vector<uint64_t> key(n);
      for (uint32_t c = 0; c < n; ++c)
        {
          int p[3];
          ijk(c, p);
          key[c] = spread(p[0]) | (spread(p[1]) << 1) | (spread(p[2]) << 2);
        }
      std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return key[x] < key[y]; });
    }
  for (uint32_t c = 0; c < n; ++c)
    position[order[c]] = c;
  for (uint32_t c = 0; c < n; ++c)
    {
      int p[3];
      ijk(order[c], p);
      if (cell_ijk)
        for (int d = 0; d < 3; ++d)
          cell_ijk[(size_t)c * 3 + d] = p[d];
      const uint64_t stride[3] = {1, (uint64_t)cells[0], (uint64_t)cells[0] * cells[1]};
      for (int d = 0; d < 3; ++d)
        {
          neighbours[(size_t)c * 6 + 2 * d] = p[d] > 0 ? (int32_t)position[order[c] - stride[d]] : MGX_DG_BOUNDARY;
          neighbours[(size_t)c * 6 + 2 * d + 1] =
            p[d] + 1 < cells[d] ? (int32_t)position[order[c] + stride[d]] : MGX_DG_BOUNDARY;
        }
    }
  return MGX_OK;
}

} // extern "C"

/* ---------------------------------------------------------------------------------------------
 * MultigridSolverDG
 * --------------------------------------------------------------------------------------------- */
namespace
{
  size_t dg_nsz(int number) { return number == MGX_F64 ? 8 : 4; }

  template <typename T>
  __global__ void __launch_bounds__(256)
    k_start_vector(T *__restrict__ v, const uint32_t *__restrict__ cell_id, uint32_t n_cells, uint32_t n3, double mean)
  {
    const uint64_t total = (uint64_t)n_cells * n3;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x)
      {
        const uint32_t c = (uint32_t)(i / n3);
        const uint64_t g = (uint64_t)(cell_id ? cell_id[c] : c) * n3 + (i - (uint64_t)c * n3);
        v[i]             = (T)((double)(g % 11u) - mean);
      }
  }

  int dg_dot(mgx_dg_solver_t S, int number, const void *x, const void *y, double *out)
  {
    return mgx::dot_owned_prefix(S->ctx, number, x, y, S->n, out);
  }

  int dg_norm(mgx_dg_solver_t S, int number, const void *x, double *out)
  {
    MGX_DG_TRY(mgx::dot_owned_prefix(S->ctx, number, x, x, S->n, out));
    *out = std::sqrt(*out);
    return MGX_OK;
  }

  // PreconditionChebyshev<LaplaceOperatorCompactCombine, Vector, JacobiTransformed>: vmult (zero start) and
  // step, through the merged operation (deal.II hands iteration index 0 / 1, then k + 1 / k + 2)
  int dg_smoother_apply(mgx_dg_solver_t S, bool is_step)
  {
    const mgx_smoother_info &I = S->info;
    int                      index;
    if (!is_step)
      {
        MGX_DG_TRY(mgx_dg_vmult_with_chebyshev_update(S->A, S->defect, 0, 0., 1. / I.theta, S->update, S->old));
        index = 1;
      }
    else
      {
        MGX_DG_TRY(mgx_dg_vmult_with_chebyshev_update(S->A, S->defect, 1, 0., 1. / I.theta, S->update, S->old));
        std::swap(S->update, S->old);
        index = 2;
      }
    if (I.degree < 2 || std::fabs(I.delta) < 1e-40)
      return MGX_OK;
    double rhok = I.delta / I.theta;
    const double sigma = I.theta / I.delta;
    for (int k = 0; k < I.degree - 1; ++k, ++index)
      {
        const double rhokp = 1. / (2. * sigma - rhok);
        const double f1 = rhokp * rhok, f2 = 2. * rhokp / I.delta;
        rhok = rhokp;
        MGX_DG_TRY(mgx_dg_vmult_with_chebyshev_update(S->A, S->defect, (unsigned)index, f1, f2, S->update, S->old));
        std::swap(S->update, S->old);
      }
    return MGX_OK;
  }

  // vmult_residual_and_restrict_to_cg (laplace_operator_dg.h:853-861, action 1 :1798-1819): cg = sum over the cells of
  // P^T (rhs - A lhs), inside the cell kernel.  Cells c, c + 8, ... share no FE_Q DoF when the mesh is in forest order:
  // eight launches with plain adds (the sum is then the same in every run); one launch with atomics otherwise.
  int dg_residual_and_restrict(mgx_dg_solver_t S, void *cg, const void *rhs, const void *lhs)
  {
    hipStream_t       s  = (hipStream_t)mgx_context_stream(S->ctx);
    mgx_dg_operator_t op = S->A;
    DG_HIP(hipMemsetAsync(cg, 0, dg_nsz(S->number) * S->n_cg, s));
    if (op->n_ghost > 0)
      MGX_DG_TRY(update_ghosts(op, const_cast<void *>(lhs)));
    MergedArgs m;
    m.cg    = cg;
    m.idx27 = S->idx27;
    m.P1    = S->P1;
    if (!S->cg_eight_colours)
      MGX_DG_TRY(launch_cells(op, kRestrict, nullptr, rhs, lhs, 0, 0, 0, nullptr, op->n_cells, nullptr, 0, m));
    else
      {
        m.plain       = 1;
        m.cell_stride = 8;
        for (uint32_t k = 0; k < 8 && k < op->n_cells; ++k)
          MGX_DG_TRY(launch_cells(op, kRestrict, nullptr, rhs, lhs, 0, 0, 0, nullptr, (op->n_cells - k + 7) / 8, nullptr, k, m));
      }
    if (S->decomposed) // FE_Q DoFs on a rank interface collect the contributions of all sharers
      MGX_DG_TRY(mgx_exchange_add(S->fe, cg));
    return MGX_OK;
  }

  // dg_v_cycle(1) (multigrid_solver_dg.h:605-633): defect in, update out
  int dg_v_cycle(mgx_dg_solver_t S)
  {
    hipStream_t s = (hipStream_t)mgx_context_stream(S->ctx);
    MGX_DG_TRY(dg_smoother_apply(S, false));
    // vmult_residual_and_restrict_to_cg (:616-618)
    if (!mgx::context_tunables(S->ctx).dg_unmerged_restrict)
      MGX_DG_TRY(dg_residual_and_restrict(S, S->cg_defect, S->defect, S->update));
    else
      {
        MGX_DG_TRY(mgx_dg_vmult_residual(S->A, S->t, S->defect, S->update));
        DG_HIP(hipMemsetAsync(S->cg_defect, 0, dg_nsz(S->number) * S->n_cg, s));
        mgx::launch_dg_cg_transfer(s, S->number, S->degree, false, S->cg_defect, S->t, S->idx27, S->n_cells, S->P1,
                                   S->cg_eight_colours);
        if (S->decomposed) // FE_Q DoFs on a rank interface collect the contributions of all sharers
          MGX_DG_TRY(mgx_exchange_add(S->fe, S->cg_defect));
      }
    MGX_DG_TRY(mgx_solver_v_cycle(S->cfe)); // :622
    // prolongate_add_cg_to_dg (:625; laplace_operator_dg.h:1863-1894)
    mgx::launch_dg_cg_transfer(s, S->number, S->degree, true, S->update, S->cg_update, S->idx27, S->n_cells, S->P1);
    DG_HIP(hipGetLastError());
    return dg_smoother_apply(S, true); // :629
  }
} // namespace

extern "C" {

int mgx_dg_solver_create(mgx_context_t ctx, const mgx_dg_solver_desc *desc, mgx_dg_solver_t *out)
{
  if (!ctx || !desc || !out || !desc->matrix_dg || !desc->matrix_dg_dp || !desc->cfe)
    return dg_fail(MGX_ERR_INVALID_ARGUMENT, "mgx_dg_solver_create: null argument");
  mgx_dg_operator_t A = desc->matrix_dg, Ad = desc->matrix_dg_dp;
  if (Ad->number != MGX_F64 || A->n_cells != Ad->n_cells || A->degree != Ad->degree || A->basis != Ad->basis)
    return dg_fail(MGX_ERR_INVALID_ARGUMENT, "mgx_dg_solver_create: matrix_dg_dp must be the fp64 twin of matrix_dg");
  if (desc->degree_pre < 1)
    return dg_fail(MGX_ERR_INVALID_ARGUMENT, "mgx_dg_solver_create: degree_pre must be at least 1");
  const int      lmax = mgx_solver_n_levels(desc->cfe) - 1;
  mgx_operator_t fe   = nullptr;
  MGX_DG_TRY(mgx_solver_get_operator(desc->cfe, lmax, 0, &fe));
  const uint32_t *idx27 = nullptr;
  uint32_t        nc = 0, ncg = 0;
  int             p = 0;
  MGX_DG_TRY(mgx_operator_device_indices(fe, &idx27, &nc, &ncg, &p));
  if (nc != A->n_cells || p != A->degree || mgx_operator_number(fe) != A->number)
    return dg_fail(MGX_ERR_INVALID_ARGUMENT, "mgx_dg_solver_create: the FE_Q hierarchy's finest level must have the DG "
                                             "operator's cells, degree and number type");
  std::unique_ptr<mgx_dg_solver_s, int (*)(mgx_dg_solver_t)> S(new mgx_dg_solver_s, mgx_dg_solver_destroy);
  S->ctx = ctx;
  S->A = A;
  S->A_dp = Ad;
  S->cfe = desc->cfe;
  S->degree = A->degree;
  S->number = A->number;
  S->n = (size_t)mgx_dg_operator_n_dofs(A);
  S->n_vec = (size_t)mgx_dg_operator_vector_size(A);
  S->fe = fe;
  S->decomposed = A->n_ghost > 0;
  if (A->n_ghost != Ad->n_ghost)
    return dg_fail(MGX_ERR_INVALID_ARGUMENT, "mgx_dg_solver_create: the two DG operators must share the partition");
  S->idx27 = idx27;
  S->n_cells = nc;
  S->n_cg = ncg;
  {
    // forest order (the same child of every parent in one class): checked, not assumed
    std::vector<uint32_t> h27(27 * (size_t)nc), stamp(ncg, 0xFFFFFFFFu);
    DG_HIP(hipMemcpy(h27.data(), idx27, sizeof(uint32_t) * h27.size(), hipMemcpyDeviceToHost));
    bool ok = nc >= 64;
    for (uint32_t k = 0; k < 8 && ok; ++k)
      for (uint32_t c = k; c < nc && ok; c += 8)
        for (int e = 0; e < 27 && ok; ++e)
          {
            const uint32_t v = h27[27 * (size_t)c + e];
            if (v == 0xFFFFFFFFu || v >= ncg)
              continue;
            if (S->degree == 1 && (e % 3 == 1 || (e / 3) % 3 == 1 || e / 9 == 1)) // entity without DoFs
              continue;
            // an entity's first DoF identifies it; stamp = class * nc + cell would overflow: class and cell apart
            const uint32_t mark = k * 0x10000000u + (c >> 3);
            ok                  = stamp[v] == 0xFFFFFFFFu || (stamp[v] >> 28) != k || stamp[v] == mark;
            stamp[v]            = mark;
          }
    S->cg_eight_colours = ok && nc < 0x80000000u;
  }
  hipStream_t  s  = (hipStream_t)mgx_context_stream(ctx);
  const size_t vb = dg_nsz(S->number) * S->n_vec;
  for (void **v : {&S->defect, &S->t, &S->update, &S->old})
    {
      DG_HIP(hipMalloc(v, vb));
      DG_HIP(hipMemsetAsync(*v, 0, vb, s));
    }
  for (double **v : {&S->r, &S->z, &S->d, &S->h})
    {
      DG_HIP(hipMalloc((void **)v, 8 * S->n_vec));
      DG_HIP(hipMemsetAsync(*v, 0, 8 * S->n_vec, s));
    }
  {
    const int n1 = S->degree + 1;
    DG_HIP(hipMalloc(&S->P1, dg_nsz(S->number) * n1 * n1));
    if (S->number == MGX_F64)
      DG_HIP(hipMemcpy(S->P1, A->h.P1.data(), 8 * n1 * n1, hipMemcpyHostToDevice));
    else
      {
        std::vector<float> pf(A->h.P1.begin(), A->h.P1.end());
        DG_HIP(hipMemcpy(S->P1, pf.data(), 4 * n1 * n1, hipMemcpyHostToDevice));
      }
  }
  // the FE_Q hierarchy under a DG level smooths with degree_pre - 1 on its finest level and solves
  // the coarsest one to 2e-3 (multigrid_solver_dg.h:271-291)
  if (lmax > 0)
    MGX_DG_TRY(mgx_solver_reset_smoother(desc->cfe, lmax, 20., std::max(1, desc->degree_pre - 1), 15));
  {
    uint32_t n0 = 0;
    mgx_operator_t f0 = nullptr;
    MGX_DG_TRY(mgx_solver_get_operator(desc->cfe, 0, 0, &f0));
    MGX_DG_TRY(mgx_operator_device_indices(f0, nullptr, nullptr, &n0, nullptr));
    MGX_DG_TRY(mgx_solver_reset_smoother(desc->cfe, 0, 2e-3, -1, (int)std::max<uint32_t>(3u, n0)));
  }
  MGX_DG_TRY(mgx_solver_get_vector(desc->cfe, lmax, 2, &S->cg_defect));
  MGX_DG_TRY(mgx_solver_get_vector(desc->cfe, lmax, 4, &S->cg_update));

  // smooth_dg.initialize (multigrid_solver_dg.h:293-303): eigenvalue estimate by 15 iterations of
  // CG preconditioned with JacobiTransformed on v_i = (i mod 11) - mean, lambda_max = 1.2 x the
  // largest Ritz value, range 20
  {
    const size_t n  = S->n;
    void        *r = S->t, *z = S->update, *d = S->old, *h = S->defect; // free until the first cycle
    double ng = (double)n; // global number of DoFs
    MGX_DG_TRY(mgx::allreduce_sum(ctx, &ng, 1));
    const uint64_t ngl  = (uint64_t)(ng + 0.5);
    const uint64_t full = ngl / 11, rem = ngl % 11;
    const double   mean = (full * 55.0 + rem * (rem - 1) / 2.0) / (double)ngl;
    uint32_t      *cell_id = nullptr;
    if (desc->cell_global_id)
      {
        DG_HIP(hipMalloc((void **)&cell_id, sizeof(uint32_t) * (size_t)S->n_cells));
        DG_HIP(hipMemcpy(cell_id, desc->cell_global_id, sizeof(uint32_t) * (size_t)S->n_cells, hipMemcpyHostToDevice));
      }
    {
      const uint32_t n3   = (uint32_t)(n / S->n_cells);
      const uint32_t grid = (uint32_t)std::min<uint64_t>((n + 255) / 256, 8192);
      if (S->number == MGX_F64)
        hipLaunchKernelGGL(k_start_vector<double>, dim3(grid), dim3(256), 0, s, (double *)r, cell_id, S->n_cells, n3, mean);
      else
        hipLaunchKernelGGL(k_start_vector<float>, dim3(grid), dim3(256), 0, s, (float *)r, cell_id, S->n_cells, n3, mean);
      DG_HIP(hipStreamSynchronize(s));
      (void)hipFree(cell_id);
    }
    std::vector<double> diag, off;
    double              res = 0, rz = 0, rz_old = 0, alpha = 0, alpha_old = 0, beta = 0;
    MGX_DG_TRY(dg_norm(S.get(), S->number, r, &res));
    int it = 0;
    while (it < 15 && res > 1e-10)
      {
        ++it;
        rz_old = rz;
        MGX_DG_TRY(mgx_dg_jacobi_vmult(A, z, r));
        MGX_DG_TRY(dg_dot(S.get(), S->number, r, z, &rz));
        if (it > 1)
          {
            beta = rz / rz_old;
            MGX_DG_TRY(mgx_sadd(ctx, S->number, d, beta, 1.0, z, n));
          }
        else
          MGX_DG_TRY(mgx_copy_cast(ctx, d, S->number, z, S->number, n));
        alpha_old = alpha;
        MGX_DG_TRY(mgx_dg_vmult(A, h, d));
        double dh = 0;
        MGX_DG_TRY(dg_dot(S.get(), S->number, d, h, &dh));
        alpha = rz / dh;
        MGX_DG_TRY(mgx_sadd(ctx, S->number, r, 1.0, -alpha, h, n));
        MGX_DG_TRY(dg_norm(S.get(), S->number, r, &res));
        if (it == 1)
          diag.push_back(1. / alpha);
        else
          {
            off.push_back(std::sqrt(beta) / alpha_old);
            diag.push_back(1. / alpha + beta / alpha_old);
          }
      }
    mgx_smoother_info &I = S->info;
    I.cg_iterations      = it;
    if (diag.empty())
      I.lambda_min = I.lambda_max = 1.;
    else
      {
        const int           m = (int)diag.size();
        std::vector<double> Tm((size_t)m * m, 0.0), lam, V;
        for (int i = 0; i < m; ++i)
          {
            Tm[i * m + i] = diag[i];
            if (i + 1 < m)
              Tm[i * m + i + 1] = Tm[(i + 1) * m + i] = off[i];
          }
        sym_eig(m, Tm, lam, V);
        I.lambda_min = lam.front();
        I.lambda_max = 1.2 * lam.back();
      }
    const double a = I.lambda_max / 20.;
    I.degree       = desc->degree_pre;
    I.delta        = (I.lambda_max - a) * 0.5;
    I.theta        = (I.lambda_max + a) * 0.5;
    for (void *v : {S->defect, S->t, S->update, S->old})
      DG_HIP(hipMemsetAsync(v, 0, vb, s));
    DG_HIP(hipStreamSynchronize(s));
  }
  *out = S.release();
  return MGX_OK;
}

int mgx_dg_solver_destroy(mgx_dg_solver_t S)
{
  if (!S)
    return MGX_OK;
  if (S->ctx)
    (void)hipStreamSynchronize((hipStream_t)mgx_context_stream(S->ctx));
  for (void *v : {S->defect, S->t, S->update, S->old, S->P1, (void *)S->r, (void *)S->z, (void *)S->d, (void *)S->h})
    (void)hipFree(v);
  delete S;
  return MGX_OK;
}

int mgx_dg_solver_smoother_info(mgx_dg_solver_t S, mgx_smoother_info *info)
{
  if (!S || !info)
    return dg_fail(MGX_ERR_INVALID_ARGUMENT, "mgx_dg_solver_smoother_info: null argument");
  *info = S->info;
  return MGX_OK;
}

int mgx_dg_restrict_to_cg(mgx_dg_solver_t S, void *cg_dst, const void *dg_src)
{
  if (!S || !cg_dst || !dg_src)
    return dg_fail(MGX_ERR_INVALID_ARGUMENT, "mgx_dg_restrict_to_cg: null argument");
  hipStream_t s = (hipStream_t)mgx_context_stream(S->ctx);
  DG_HIP(hipMemsetAsync(cg_dst, 0, dg_nsz(S->number) * S->n_cg, s));
  mgx::launch_dg_cg_transfer(s, S->number, S->degree, false, cg_dst, dg_src, S->idx27, S->n_cells, S->P1, S->cg_eight_colours);
  DG_HIP(hipGetLastError());
  return MGX_OK;
}

int mgx_dg_vmult_residual_and_restrict_to_cg(mgx_dg_solver_t S, void *cg_dst, const void *rhs, const void *lhs)
{
  if (!S || !cg_dst || !rhs || !lhs)
    return dg_fail(MGX_ERR_INVALID_ARGUMENT, "mgx_dg_vmult_residual_and_restrict_to_cg: null argument");
  return dg_residual_and_restrict(S, cg_dst, rhs, lhs);
}

int mgx_dg_prolongate_add_cg_to_dg(mgx_dg_solver_t S, void *dg_dst, const void *cg_src)
{
  if (!S || !dg_dst || !cg_src)
    return dg_fail(MGX_ERR_INVALID_ARGUMENT, "mgx_dg_prolongate_add_cg_to_dg: null argument");
  hipStream_t s = (hipStream_t)mgx_context_stream(S->ctx);
  mgx::launch_dg_cg_transfer(s, S->number, S->degree, true, dg_dst, cg_src, S->idx27, S->n_cells, S->P1);
  DG_HIP(hipGetLastError());
  return MGX_OK;
}

int mgx_dg_solver_vmult(mgx_dg_solver_t S, double *dst, const double *src)
{
  if (!S || !dst || !src)
    return dg_fail(MGX_ERR_INVALID_ARGUMENT, "mgx_dg_solver_vmult: null argument");
  MGX_DG_TRY(mgx_copy_cast(S->ctx, S->defect, S->number, src, MGX_F64, S->n)); // multigrid_solver_dg.h:433
  MGX_DG_TRY(dg_v_cycle(S));
  return mgx_copy_cast(S->ctx, dst, MGX_F64, S->update, S->number, S->n);       // :436
}

int mgx_dg_solver_solve_cg(mgx_dg_solver_t S, double tolerance, const double *rhs, double *solution,
                           unsigned *iterations, double *reduction_rate)
{
  if (!S || !rhs || !solution)
    return dg_fail(MGX_ERR_INVALID_ARGUMENT, "mgx_dg_solver_solve_cg: null argument");
  // SolverCG with ReductionControl(100, 1e-16, tolerance), zero start, preconditioner = one DG V-cycle
  // (multigrid_solver_dg.h:410-424)
  mgx_context_t ctx = S->ctx;
  const size_t  n   = S->n;
  hipStream_t   s   = (hipStream_t)mgx_context_stream(ctx);
  double       *r = S->r, *z = S->z, *d = S->d, *h = S->h;
  DG_HIP(hipMemsetAsync(solution, 0, 8 * S->n_vec, s));
  MGX_DG_TRY(mgx_copy_cast(ctx, r, MGX_F64, rhs, MGX_F64, n));
  double res0 = 0, res = 0, rz = 0, rz_old = 0;
  MGX_DG_TRY(dg_norm(S, MGX_F64, r, &res0));
  res         = res0;
  unsigned it = 0;
  while (res > std::max(1e-16, tolerance * res0) && it < 100)
    {
      ++it;
      MGX_DG_TRY(mgx_dg_solver_vmult(S, z, r));
      rz_old = rz;
      MGX_DG_TRY(dg_dot(S, MGX_F64, r, z, &rz));
      if (it > 1)
        MGX_DG_TRY(mgx_sadd(ctx, MGX_F64, d, rz / rz_old, 1.0, z, n));
      else
        MGX_DG_TRY(mgx_copy_cast(ctx, d, MGX_F64, z, MGX_F64, n));
      MGX_DG_TRY(mgx_dg_vmult(S->A_dp, h, d));
      double dh = 0;
      MGX_DG_TRY(dg_dot(S, MGX_F64, d, h, &dh));
      const double alpha = rz / dh;
      MGX_DG_TRY(mgx_sadd(ctx, MGX_F64, solution, 1.0, alpha, d, n));
      MGX_DG_TRY(mgx_sadd(ctx, MGX_F64, r, 1.0, -alpha, h, n));
      MGX_DG_TRY(dg_norm(S, MGX_F64, r, &res));
    }
  if (iterations)
    *iterations = it;
  if (reduction_rate)
    *reduction_rate = it ? std::pow(res / res0, 1.0 / it) : 1.0;
  return res > std::max(1e-16, tolerance * res0) ? dg_fail(MGX_ERR_NOT_CONVERGED, "mgx_dg_solver_solve_cg: 100 iterations")
                                                 : MGX_OK;
}

} // extern "C"
