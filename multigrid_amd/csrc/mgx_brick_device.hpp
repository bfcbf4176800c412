// mgx_brick_device.hpp -- device code shared by the two forms of the brick cell loop
// (mgx_brick.hip: cell-by-cell rounds; mgx_macro.hip: macro-element sweeps over the assembled brick).
#pragma once

#include "mgx_internal.hpp"

#include <hip/hip_runtime.h>

namespace mgx
{
  // Brick shape: 4x4x4 cells for p <= 4 (64 consecutive Morton cells), 2x2x2 for p >= 5 (the 8
  // children of one parent), so that the (NB p + 1)^3 fp64 accumulator stays below 55 kB of LDS.
  // p <= 4: a cell needs (p+1)^2 <= 25 threads, a wave owns two cells of a round, four waves
  // process the 8 same-parity cells of a round.  p >= 5: the 8 cells of a brick are mutually
  // adjacent; two of them are integrated side by side on (p+1)^2 threads each (one or two waves
  // per cell) and added to the accumulator one after the other (fixed order: deterministic).
  template <int P, bool WIDE = false>
  struct BCfg
  {
    static constexpr int  NB         = P <= 4 ? 4 : 2; // cells per direction
    static constexpr int  NCELLS     = NB * NB * NB;
    static constexpr int  N          = P + 1;
    static constexpr int  LN         = N | 1;          // odd x-line pitch (bank-conflict free)
    static constexpr int  G          = NB * P + 1;     // points per direction
    static constexpr int  NE1        = 2 * NB + 1;     // mesh entities per direction
    static constexpr int  NE         = NE1 * NE1 * NE1;
    static constexpr int  TPC        = N * N;
    static constexpr bool kTwoPerWave = P <= 4;        // two cells per wave, 8 cells per round
    static constexpr int  TPW        = kTwoPerWave ? 32 : ((TPC + 63) / 64) * 64; // threads reserved per cell
    // WIDE (p <= 4 only): two parity classes of 8 cells side by side on 512 threads, added to the
    // accumulator one after the other.  Halves the number of sequential rounds of a workgroup: used
    // for launches with too few bricks to fill the chip, whose duration is one workgroup's latency.
    static constexpr bool kWide      = WIDE && kTwoPerWave;
    static constexpr int  ROUND_CELLS = kTwoPerWave ? (kWide ? 16 : 8) : 2;
    static constexpr int  ROUNDS     = NCELLS / ROUND_CELLS;
    static constexpr int  THREADS    = kTwoPerWave ? (kWide ? 512 : 256) : 2 * TPW;
    static constexpr bool kWaveSync  = kTwoPerWave || TPW == 64; // transposes stay inside one wave
    static constexpr int  CELL_LDS   = N * N * LN;
  };

  // fused post-operations (what the reference passes as operation_after_loop)
  enum BrickMode
  {
    kPlain    = 0, // out = A src                                   (vmult, laplace_operator.h:573)
    kResidual = 1, // out = a - A src                               (vmult_residual, :605)
    kCheb     = 2, // out = x + f1 (x - out) + f2 b (a - A x)       (PreconditionChebyshev update)
    kChebFirst = 3, // out = x + f2 b (a - A x)                     (first step: no x_old term)
    kChebZeroOld = 4, // out = x + f1 x + f2 b (a - A x)             (x_old known to be zero)
    // start of PreconditionChebyshev::vmult (zero initial guess): the first iterate x_1 = f0 b a is
    // never stored -- the first loop iteration computes it while gathering (kChebInit, x_old = 0),
    // the second one recomputes it as its x_old (kChebOldInit); separable kernel only
    kChebInit    = 5, // x := f0 b a ; out = x + f1 x + f2 b (a - A x)
    kChebOldInit = 6, // out = x + f1 (x - f0 b a) + f2 b (a - A x)
    // V-cycle: the residual a - A x is only needed restricted to the next coarser level
    // (multigrid_solver.h:663-668).  Every brick restricts the residual values it completes (its
    // LAST points, everything else masked to zero) with the transposed embedding and adds the
    // (PB p + 1)^3 coarse values to the coarse vector; the residual itself is never stored.
    kResidualRestrict = 7,
    // fused PCG step (vmult_with_cg_update, laplace_operator.h:638-719; macro-element kernel only):
    // the gather forms p_new = f2 p + q (f1 == 0: p_new = q); at completion x += f1 p_old,
    // p = p_new, q = A p_new, and q.p, r.r, q.r, q.q are accumulated per workgroup
    kCgUpdate = 8,
    // first post-smoothing iteration of the V-cycle with the coarse-grid correction formed on the
    // fly (multigrid_solver.h:674-678; macro-element kernel only): the gather adds the prolongated
    // coarse values to x (prolong_brick), the iteration is kChebFirst on the corrected x, which is
    // also written back at completion (it is x_old of the next iteration).  coarse / coarse_blocks
    // as for kResidualRestrict.
    kChebFirstProlong = 9
  };

  template <typename T>
  struct BrickPost
  {
    const T *a;       // kResidual: rhs ; kCheb: rhs b of the smoother
    const T *b;       // kCheb: inverse diagonal
    const T *old;     // kCheb: previous iterate x_old (may alias out: read before written)
    T       *out;     // result vector
    T       *partial; // carrier of partial sums between colour launches (may alias out)
    T        f1, f2, f0;
    T              *coarse;        // kResidualRestrict: coarse-level vector the restriction adds to
    const uint32_t *coarse_blocks; // kResidualRestrict: coarse entity table of the brick's parents
    T              *coarse_scratch; // kResidualRestrict: [brick][CN^3] restricted values per brick (nullptr: added into `coarse`)
    // kCgUpdate: a = r, b = q (second gathered operand), old = x, out = q, f1 = alpha, f2 = beta
    T              *src_w;         // kCgUpdate: the source vector p, written at completion
    T              *x_w;           // kCgUpdate: x, updated at completion
    double         *sums;          // kCgUpdate: [gridDim.x * 4] partial sums of this launch
    // reduced-colour schedules (mgx_macro.hip, FREE): priv = the bricks' blocks of private values,
    // [brick][n_surf], priv_bytes in all; surf_off[slot] = offset of the private entity `slot` inside a
    // block (0xFFFFFFFF: not private, the entity follows its FIRST / LAST flags)
    T              *priv;
    const uint32_t *surf_off;
    uint32_t        n_surf, priv_bytes;
  };

  // Entity table word: bits 0..29 first DoF of the entity, bit 30 FIRST, bit 31 LAST;
  // 0xFFFFFFFF = constrained / empty entity (the host refuses levels with >= 2^30 - 1 DoFs)
  __device__ __forceinline__ uint32_t ent_index(uint32_t w) { return w & 0x3FFFFFFFu; }
  __device__ __forceinline__ uint32_t ent_flags_of(uint32_t w) { return w >> 30; }

  // wave-local ordering of LDS traffic: the two cells of a wave exchange data only among the
  // lanes of that wave, which execute in lockstep; the LDS services one wave's operations in
  // order, so a compiler-level barrier is all that is needed between the transposes
  __device__ __forceinline__ void wave_sync()
  {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
  }

  // workgroup barrier that orders LDS traffic only: global loads issued earlier (the prefetch of
  // the next round's source values) stay in flight across it, which __syncthreads() would drain
  __device__ __forceinline__ void lds_barrier()
  {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }

#ifndef MGX_PACKED_F32
#define MGX_PACKED_F32 1
#endif
  template <int N, typename T>
  __device__ __forceinline__ void eo_split(const T (&x)[N], T (&xe)[N / 2 + 1], T (&xo)[N / 2 + 1])
  {
    constexpr int H = N / 2;
#pragma unroll
    for (int i = 0; i < H; ++i)
      {
        xe[i] = x[i] + x[N - 1 - i];
        xo[i] = x[i] - x[N - 1 - i];
      }
    xe[H] = (N % 2) ? x[H] : T(0); // middle entry
    xo[H] = T(0);
  }

  template <int N, typename T>
  __device__ __forceinline__ void eo_apply(const EOMat<T> &A, const T (&xe)[N / 2 + 1], const T (&xo)[N / 2 + 1],
                                           T (&y)[N])
  {
    constexpr int H = N / 2;
    if constexpr (sizeof(T) == 4 && MGX_PACKED_F32)
      {
        // fp32: the even and the odd half-products are two independent sums of the same length -- one stream
        // of 2-vector multiply-adds (v_pk_fma_f32), half the instructions of the sweeps that bound this form
        typedef T T2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int a = 0; a < H; ++a)
          {
            T2 r = T2{A.ee[a * H], A.eo[a * H]} * T2{xe[0], xo[0]};
#pragma unroll
            for (int i = 1; i < H; ++i)
              r = __builtin_elementwise_fma(T2{A.ee[a * H + i], A.eo[a * H + i]}, T2{xe[i], xo[i]}, r);
            if (N % 2)
              r[0] = fma(A.mc[a], xe[H], r[0]);
            y[a]         = r[0] + r[1];
            y[N - 1 - a] = r[0] - r[1];
          }
        if (N % 2)
          {
            T r = A.mhh * xe[H];
#pragma unroll
            for (int i = 0; i < H; ++i)
              r = fma(A.mc[i], xe[i], r);
            y[H] = r;
          }
        return;
      }
#pragma unroll
    for (int a = 0; a < H; ++a)
      {
        T r0 = A.ee[a * H] * xe[0];
        T r1 = A.eo[a * H] * xo[0];
#pragma unroll
        for (int i = 1; i < H; ++i)
          {
            r0 = fma(A.ee[a * H + i], xe[i], r0);
            r1 = fma(A.eo[a * H + i], xo[i], r1);
          }
        if (N % 2)
          r0 = fma(A.mc[a], xe[H], r0);
        y[a]         = r0 + r1;
        y[N - 1 - a] = r0 - r1;
      }
    if (N % 2)
      {
        T r = A.mhh * xe[H];
#pragma unroll
        for (int i = 0; i < H; ++i)
          r = fma(A.mc[i], xe[i], r);
        y[H] = r;
      }
  }

  // ------------------------------------------------------------------------------------------
  // Restriction of the brick array in place (kResidualRestrict).  acc holds the G^3 fine values of
  // the brick (G = NB p + 1) whose cells are the children of PB^3 parents (PB = NB / 2): three 1D
  // sweeps with the transposed embedding P1 (one line per thread, values in registers, outputs
  // written over the head of the line), then the (PB p + 1)^3 coarse values are added to the
  // coarse vector through the coarse entity table of the brick.  Bricks of one colour launch are
  // not adjacent, so their parents share no coarse DoF: plain read-modify-write.
  // ------------------------------------------------------------------------------------------
  // Scalar operands of a transfer line product in slices: called after row / column k of an M x N block of the embedding
  // whose rows (columns) hold ROW coefficients, it keeps the loads of the next slice behind the products of this one, so
  // that no more than ~40 doubles of coefficients are live at a time.  Without it the compiler loads the whole block
  // ahead, overflows the scalar register file (~100 registers, two per double) and spills it lane by lane into vector
  // registers: round 4 found three v_readlane per multiply-add in the p = 8 kernels (5240 + 1670 lane moves among the
  // 9500 vector instructions per brick of the residual + restriction form).  Blocks of up to 48 coefficients fit.
  template <int ROW, int TOTAL>
  __device__ __forceinline__ void scalar_operand_slice(int k)
  {
    constexpr int C = (40 / ROW) < 1 ? 1 : 40 / ROW;
    if (TOTAL > 48 && (k + 1) % C == 0)
      {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
  }

  // one parent's part of a line: o[j] = sum_a P1[a][j] r[a], a over the 2p+1 fine points of parent pb.
  // Even-odd form (pe = Basis1D::P1eo): the sums and differences of the fine values at mirrored positions meet the
  // symmetric and the antisymmetric half of the embedding -- (p+1) nh + p nh (+ p+1) multiply-adds instead of
  // (2p+1)(p+1): 23 for 45 at p = 4, 77 for 153 at p = 8, and half as many scalar operands.
  template <int P, typename T>
  __device__ __forceinline__ void restrict_half(const T *__restrict__ pe, const T (&r)[2 * P + 1], T (&o)[P + 1])
  {
    constexpr int NH = (P + 1) / 2, TOT = (2 * P + 1) * NH + ((P % 2 == 0) ? P + 1 : 0);
#ifdef MGX_RESTRICT_DENSE // A/B build: the dense products with P1 itself
    {
      constexpr int N = P + 1, M = 2 * P + 1;
      const T *p1 = reinterpret_cast<const Basis1D<T> *>(reinterpret_cast<const char *>(pe) - offsetof(Basis1D<T>, P1eo))->P1;
#pragma unroll
      for (int j = 0; j < N; ++j)
        {
          T s = p1[j] * r[0];
#pragma unroll
          for (int a = 1; a < M; ++a)
            s = fma(p1[a * N + j], r[a], s);
          o[j] = s;
          scalar_operand_slice<M, M * N>(j);
        }
      return;
    }
#endif
    const T      *he = pe, *ho = pe + (P + 1) * NH, *pc = pe + (2 * P + 1) * NH;
    T             re[P + 1], ro[P];
#pragma unroll
    for (int a = 0; a < P; ++a)
      {
        re[a] = r[a] + r[2 * P - a];
        ro[a] = r[a] - r[2 * P - a];
      }
    re[P] = r[P];
#pragma unroll
    for (int j = 0; j < NH; ++j)
      {
        T se = he[j] * re[0], so = ho[j] * ro[0];
#pragma unroll
        for (int a = 1; a <= P; ++a)
          se = fma(he[a * NH + j], re[a], se);
#pragma unroll
        for (int a = 1; a < P; ++a)
          so = fma(ho[a * NH + j], ro[a], so);
        o[j]     = se + so;
        o[P - j] = se - so;
        scalar_operand_slice<2 * P + 1, TOT>(j);
      }
    if (P % 2 == 0)
      {
        T s = pc[0] * re[0];
#pragma unroll
        for (int a = 1; a <= P; ++a)
          s = fma(pc[a], re[a], s);
        o[P / 2] = s;
      }
  }

  // Work items of a sweep are HALF lines (one parent's 2p+1 fine points -> p+1 coarse values) where a line
  // spans two parents (p <= 4), two neighbouring lanes per line: 289 + 153 + 81 lines on 256 threads take
  // 3 + 2 + 1 passes of half the work instead of 2 + 1 + 1 passes of the whole.  The coarse point the two
  // parents share receives both contributions (minus the fine value counted by both) from the lower lane.
  template <int P, typename T, int NT, bool SCRATCH = false>
  __device__ __forceinline__ void restrict_brick(int tid, T *acc, const T *__restrict__ p1, T *__restrict__ coarse,
                                                 const uint32_t *__restrict__ ctab, T *__restrict__ scratch = nullptr)
  {
    // SCRATCH (scratch != nullptr): the CN^3 restricted values of the brick go to scratch[(z CN + y) CN + x]
    // instead of being added into the coarse vector: no two bricks write the same address, the bricks of a level
    // need no order among each other (coarse_assemble_kernel adds the blocks up per coarse DoF afterwards)
    using C           = BCfg<P>;
    constexpr int G   = C::G, PB = C::NB / 2, CN = PB * P + 1, CE1 = 2 * PB + 1, N = P + 1, M = 2 * P + 1;
    auto layer = [](int a, int &e, int &o, int &n) {
      const int q = a / P, rr = a - q * P;
      e           = 2 * q + (rr != 0);
      o           = rr ? rr - 1 : 0;
      n           = rr ? P - 1 : 1;
    };
    // the coarse values this thread will add to (z sweep, one coarse point per parent-half item): requested
    // now, two sweeps ahead of their use
    constexpr int NZ = (CN * CN * PB + NT - 1) / NT;
    T            *cp[SCRATCH ? 1 : NZ][N];
    T             cv[SCRATCH ? 1 : NZ][N];
    bool          ok[SCRATCH ? 1 : NZ][N];
#pragma unroll
    for (int it = 0; it < (SCRATCH ? 0 : NZ); ++it)
      {
        const int t = tid + it * NT, l = t / PB, pb = t % PB;
        const bool live = t < CN * CN * PB;
        const int  x = live ? l % CN : 0, y = live ? l / CN : 0;
        int        ex, ey, ox, oy, nx, ny;
        layer(x, ex, ox, nx);
        layer(y, ey, oy, ny);
#pragma unroll
        for (int j = 0; j < N; ++j)
          {
            int ez, oz, nz;
            layer(pb * P + j, ez, oz, nz);
            const uint32_t w = ctab[(ez * CE1 + ey) * CE1 + ex];
            // the shared point of two parents is written by the lower one
            ok[it][j] = live && w != kInvalid && !(PB == 2 && pb == 1 && j == 0);
            cp[it][j] = coarse + (ok[it][j] ? w + (uint32_t)((oz * ny + oy) * nx + ox) : 0u);
            cv[it][j] = *cp[it][j];
          }
      }
    // generic sweep over `lines` lines: base(l) = first point of line l, stride between its points
    auto sweep = [&](int lines, auto base, int stride, auto sink) {
#pragma unroll 1
      for (int t0 = 0; t0 < lines * PB; t0 += NT)
        {
          const int  t = t0 + tid, l = t / PB, pb = t % PB;
          const bool live = t < lines * PB;
          T          r[M], o[N];
          const int  b0 = base(live ? l : 0) + pb * 2 * P * stride;
#pragma unroll
          for (int a = 0; a < M; ++a)
            r[a] = acc[b0 + a * stride];
          restrict_half<P, T>(p1, r, o);
          if (PB == 2)
            {
              // lower lane: its last coarse point also takes the upper lane's first (the fine value at
              // that point was counted by both)
              const T up = __shfl_down(o[0], 1);
              if (pb == 0)
                o[P] += up - r[2 * P];
            }
          sink(live, l, pb, o);
        }
    };
    // x: lines (y, z), results over the head of the line
    sweep(
      G * G, [&](int l) { return l * G; }, 1,
      [&](bool live, int l, int pb, const T(&o)[N]) {
        // (a line's two items sit in neighbouring lanes of one wave: the reads above are done)
        if (live)
          {
#pragma unroll
            for (int j = (pb == 0 ? 0 : 1); j < N; ++j)
              acc[l * G + pb * P + j] = o[j];
          }
      });
    lds_barrier(); // LDS only: the gathered source values of the next brick stay in flight
    // y: lines (x < CN, z)
    sweep(
      CN * G, [&](int l) { return (l / CN) * G * G + l % CN; }, G,
      [&](bool live, int l, int pb, const T(&o)[N]) {
        if (live)
          {
            const int x = l % CN, z = l / CN;
#pragma unroll
            for (int j = (pb == 0 ? 0 : 1); j < N; ++j)
              acc[(z * G + pb * P + j) * G + x] = o[j];
          }
      });
    lds_barrier();
    // z: lines (x, y) with x, y < CN; the results are added to the coarse vector (values requested above;
    // all stores behind all loads: a load behind a store to the same vector would wait for it)
    {
      int it = 0;
      sweep(
        CN * CN, [&](int l) { return (l / CN) * G + l % CN; }, G * G,
        [&](bool live, int l, int pb, const T(&o)[N]) {
          if (SCRATCH)
            {
              if (live)
                {
#pragma unroll
                  for (int j = (pb == 0 ? 0 : 1); j < N; ++j)
                    scratch[(size_t)((pb * P + j) * CN * CN + l)] = o[j];
                }
            }
          else
            {
#pragma unroll
              for (int k = 0; k < (SCRATCH ? 0 : NZ); ++k)
                if (k == it)
                  {
#pragma unroll
                    for (int j = 0; j < N; ++j)
                      if (ok[SCRATCH ? 0 : k][j])
                        *cp[SCRATCH ? 0 : k][j] = cv[SCRATCH ? 0 : k][j] + o[j];
                  }
            }
          ++it;
        });
    }
  }

  // ------------------------------------------------------------------------------------------
  // Prolongation of the coarse values of a brick's parents onto the brick array (kChebFirstProlong):
  // cv = the (PB p + 1)^3 coarse values in registers (thread tid holds coarse points tid + k NT),
  // acc receives P cv on the G^3 points of the brick.  Exact embedding: every fine point takes the
  // value of the coarse finite-element function (MGTransferMatrixFree::prolongate).
  // ------------------------------------------------------------------------------------------
  // one parent: f[a] = sum_i P1[a][i] c[i], a over its 2p+1 fine points, in the even-odd form of restrict_half
  template <int P, typename T>
  __device__ __forceinline__ void prolong_half(const T *__restrict__ pe, const T (&c)[P + 1], T (&f)[2 * P + 1])
  {
    constexpr int NH = (P + 1) / 2, TOT = (2 * P + 1) * NH + ((P % 2 == 0) ? P + 1 : 0);
    constexpr int ROW = 2 * NH + (P % 2 == 0 ? 1 : 0);
    const T      *he = pe, *ho = pe + (P + 1) * NH, *pc = pe + (2 * P + 1) * NH;
    T             ce[NH], co[NH];
#pragma unroll
    for (int i = 0; i < NH; ++i)
      {
        ce[i] = c[i] + c[P - i];
        co[i] = c[i] - c[P - i];
      }
    const T cc = c[P / 2];
#pragma unroll
    for (int a = 0; a <= P; ++a)
      {
        T se = he[a * NH] * ce[0];
#pragma unroll
        for (int i = 1; i < NH; ++i)
          se = fma(he[a * NH + i], ce[i], se);
        if (P % 2 == 0)
          se = fma(pc[a], cc, se);
        if (a < P)
          {
            T so = ho[a * NH] * co[0];
#pragma unroll
            for (int i = 1; i < NH; ++i)
              so = fma(ho[a * NH + i], co[i], so);
            f[a]         = se + so;
            f[2 * P - a] = se - so;
          }
        else
          f[P] = se;
        scalar_operand_slice<ROW, TOT>(a);
      }
  }

  // f = P1 c per parent, in the even-odd form of restrict_half (pe = Basis1D::P1eo): f[a] and f[2p-a] from the sums and
  // differences of the coarse values at mirrored nodes
  template <int P, typename T>
  __device__ __forceinline__ void prolong_line(const T *__restrict__ pe, const T (&c)[(BCfg<P>::NB / 2) * P + 1],
                                               T (&f)[BCfg<P>::G])
  {
    constexpr int PB = BCfg<P>::NB / 2, NH = (P + 1) / 2, TOT = (2 * P + 1) * NH + ((P % 2 == 0) ? P + 1 : 0);
    constexpr int ROW = 2 * NH + (P % 2 == 0 ? 1 : 0);
#ifdef MGX_PROLONG_DENSE // A/B build: the dense products with P1 itself
    {
      constexpr int N = P + 1, M = 2 * P + 1;
      const T *p1 = reinterpret_cast<const Basis1D<T> *>(reinterpret_cast<const char *>(pe) - offsetof(Basis1D<T>, P1eo))->P1;
#pragma unroll
      for (int pb = 0; pb < PB; ++pb)
        {
#pragma unroll
          for (int a = (pb == 0 ? 0 : 1); a < M; ++a)
            {
              T s = p1[a * N] * c[pb * P];
#pragma unroll
              for (int i = 1; i < N; ++i)
                s = fma(p1[a * N + i], c[pb * P + i], s);
              f[pb * 2 * P + a] = s;
              scalar_operand_slice<N, M * N>(a);
            }
        }
      return;
    }
#endif
    const T      *he = pe, *ho = pe + (P + 1) * NH, *pc = pe + (2 * P + 1) * NH;
#pragma unroll
    for (int pb = 0; pb < PB; ++pb)
      {
        T ce[NH], co[NH];
#pragma unroll
        for (int i = 0; i < NH; ++i)
          {
            ce[i] = c[pb * P + i] + c[pb * P + P - i];
            co[i] = c[pb * P + i] - c[pb * P + P - i];
          }
        const T cc = c[pb * P + P / 2];
#pragma unroll
        for (int a = 0; a <= P; ++a)
          {
            T se = he[a * NH] * ce[0];
#pragma unroll
            for (int i = 1; i < NH; ++i)
              se = fma(he[a * NH + i], ce[i], se);
            if (P % 2 == 0)
              se = fma(pc[a], cc, se);
            if (a < P)
              {
                T so = ho[a * NH] * co[0];
#pragma unroll
                for (int i = 1; i < NH; ++i)
                  so = fma(ho[a * NH + i], co[i], so);
                if (pb == 0 || a > 0) // the first point of the upper parent is the last of the lower one
                  f[pb * 2 * P + a] = se + so;
                f[pb * 2 * P + 2 * P - a] = se - so;
              }
            else
              f[pb * 2 * P + P] = se;
            scalar_operand_slice<ROW, TOT>(a);
          }
      }
  }

  template <int P>
  __device__ __forceinline__ void coarse_point(int l, int &slot, uint32_t &off, int &pnt)
  {
    using C           = BCfg<P>;
    constexpr int G   = C::G, PB = C::NB / 2, CN = PB * P + 1, CE1 = 2 * PB + 1;
    const int     x = l % CN, y = (l / CN) % CN, z = l / (CN * CN);
    auto          layer = [](int a, int &e, int &o, int &n) {
      const int q = a / P, rr = a - q * P;
      e           = 2 * q + (rr != 0);
      o           = rr ? rr - 1 : 0;
      n           = rr ? P - 1 : 1;
    };
    int ex, ey, ez, ox, oy, oz, nx, ny, nz;
    layer(x, ex, ox, nx);
    layer(y, ey, oy, ny);
    layer(z, ez, oz, nz);
    slot = (ez * CE1 + ey) * CE1 + ex;
    off  = (uint32_t)((oz * ny + oy) * nx + ox);
    pnt  = (z * G + y) * G + x;
  }

  // (half-line work items as in restrict_brick were measured here too: 197 -> 253 us per colour launch of the
  // fused form, the kernel spills 176 instead of 56 B per lane with them)
  template <int P, typename T, int NT>
  __device__ __forceinline__ void prolong_brick(int tid, T *acc, const T *__restrict__ p1,
                                                const T (&cv)[((BCfg<P>::NB / 2) * P + 1) * ((BCfg<P>::NB / 2) * P + 1) *
                                                                ((BCfg<P>::NB / 2) * P + 1) / NT + 1])
  {
    using C           = BCfg<P>;
    constexpr int G   = C::G, PB = C::NB / 2, CN = PB * P + 1, NC = CN * CN * CN, NCV = NC / NT + 1;
#pragma unroll
    for (int k = 0; k < NCV; ++k)
      {
        const int l = tid + k * NT;
        if (l < NC)
          {
            int      slot, pnt;
            uint32_t off;
            coarse_point<P>(l, slot, off, pnt);
            acc[pnt] = cv[k];
          }
      }
    lds_barrier(); // LDS only: the gathered source values of the brick stay in flight
    // z: lines (x, y) with x, y < CN
    for (int l = tid; l < CN * CN; l += NT)
      {
        const int x = l % CN, y = l / CN;
        T         c[CN], f[G];
#pragma unroll
        for (int i = 0; i < CN; ++i)
          c[i] = acc[(i * G + y) * G + x];
        prolong_line<P, T>(p1, c, f);
#pragma unroll
        for (int i = 0; i < G; ++i)
          acc[(i * G + y) * G + x] = f[i];
      }
    lds_barrier(); // LDS only: the gathered source values of the brick stay in flight
    // y: lines (x < CN, z)
    for (int l = tid; l < CN * G; l += NT)
      {
        const int x = l % CN, z = l / CN;
        T         c[CN], f[G];
#pragma unroll
        for (int i = 0; i < CN; ++i)
          c[i] = acc[(z * G + i) * G + x];
        prolong_line<P, T>(p1, c, f);
#pragma unroll
        for (int i = 0; i < G; ++i)
          acc[(z * G + i) * G + x] = f[i];
      }
    lds_barrier(); // LDS only: the gathered source values of the brick stay in flight
    // x: lines (y, z)
    for (int l = tid; l < G * G; l += NT)
      {
        T c[CN], f[G];
#pragma unroll
        for (int i = 0; i < CN; ++i)
          c[i] = acc[l * G + i];
        prolong_line<P, T>(p1, c, f);
#pragma unroll
        for (int i = 0; i < G; ++i)
          acc[l * G + i] = f[i];
      }
    lds_barrier(); // LDS only: the gathered source values of the brick stay in flight
  }
} // namespace mgx
