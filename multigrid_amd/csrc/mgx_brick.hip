// mgx_brick.hip -- the round-1 cell-by-cell brick loop, kept as a CROSS-CHECK of the macro-element
// kernel (mgx_macro.hip, the production cell loop): compiled only into builds with MGX_CELLS_FORM=1
// (make CELLS_FORM=1; option "cells_form" of a context), never into the default library.  What is
// always compiled from this file is launch_brick_loop, the dispatcher.
// Atomic-free, deterministic, with the vector updates of the caller fused in (the GPU counterpart of MatrixFree::cell_loop(..., operation_before_loop,
// operation_after_loop), laplace_operator.h:605-634, 723-741).
//
// Work decomposition
//   * 64 consecutive cells that form a 4x4x4 brick (two uniform refinements of a common ancestor,
//     Morton order inside) are one workgroup's job.  The brick's (4p+1)^3 result values are
//     accumulated in LDS; nothing is added to global memory with atomics.
//   * inside the brick the cells are processed in 8 rounds of 8 same-parity cells (no two cells
//     of a round share a DoF), 8 x (p+1)^2 threads per round, one 1D line per thread, sweeps in
//     registers, transposes through LDS (see mgx_kernels.hip).
//   * bricks are coloured (8 colours on a structured mesh); one launch per colour, so bricks
//     that share surface DoFs never run concurrently.  For every mesh entity of a brick the host
//     precomputes two flags: FIRST (no earlier launch touched it: store, do not read -- this is
//     the "zero dst within the loop" of laplace_operator.h:590) and LAST (no later launch will
//     touch it: the sum is complete, so the fused post-operation runs here).  Interior entities
//     are FIRST|LAST: they are written exactly once.  Surface entities carry their partial sum
//     through global memory between launches: (8 + 16(k-1)) B for an entity shared by k bricks.
//
//   * fused post-operations (BrickMode): residual, the Chebyshev updates of the smoother (incl. the
//     two forms that never store the first iterate of a zero-start sweep) and the V-cycle's
//     residual + restriction to the next coarser level in one pass (restrict_brick below).
//   * launches that do not fill the chip use a 512-thread form (two parity classes side by side);
//     p >= 5 works on 2x2x2 bricks with two cells side by side.
//
// HBM traffic per brick at p=4 (fp64, plain form, measured 374 MB per colour launch of 4096
// bricks = 91 kB per brick): 41 kB gathered source (17^3 points for 16^3 owned, brick-grouped
// numbering), 46 kB result + partial sums written/re-read on the brick surface, 2.9 kB entity
// table -- against the algorithmic 66 kB (4096 DoFs x 16 B).  DESIGN.md 4.1 has the time budget.
#include "mgx_brick_device.hpp"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <string>

#ifndef MGX_CELLS_FORM
#define MGX_CELLS_FORM 0
#endif

namespace mgx
{
#if MGX_CELLS_FORM

  template <int N, typename T>
  __device__ __forceinline__ void bmv(const T *__restrict__ M, const T (&in)[N], T (&out)[N])
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

  template <int N, typename T>
  __device__ __forceinline__ void bmvT(const T *__restrict__ M, const T (&in)[N], T (&out)[N])
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



  // ------------------------------------------------------------------------------------------
  // Write-out of the brick accumulator with the fused post-operation, one mesh-entity class at a
  // time (class = which directions are cell-interior: hexes, 3 kinds of quads, 3 kinds of lines,
  // vertices).  Consecutive work items are consecutive DoFs of one entity, i.e. consecutive
  // global addresses: runs of (p-1)^3, (p-1)^2, (p-1) or 1 values instead of the runs of p-1 a
  // lexicographic sweep over the brick points would give.
  // ------------------------------------------------------------------------------------------

  // Pass 1 of the write-out: everything that has to be READ (partial sums of earlier launches and
  // the operands of the fused post-operation) is loaded with unconditional, branch-free loads
  // (masked entries read element 0) and folded into the accumulator value.  Pass 2 then only
  // STORES.  Keeping loads and stores in separate passes matters on gfx950: vmcnt counts stores
  // too, so a load that follows stores would wait for every earlier store's acknowledgement.
  template <int P, typename T, int MODE>
  __device__ __forceinline__ T post_value(const T *__restrict__ src, const BrickPost<T> &post, bool valid,
                                          uint32_t idx, uint8_t fl, T val)
  {
    const bool     need_partial = valid && !(fl & 1);
    const bool     last         = valid && (fl & 2);
    const uint32_t ip           = need_partial ? idx : 0u;
    const uint32_t il           = last ? idx : 0u;
    const T        pv           = post.partial[ip];
    if (MODE == kPlain)
      return need_partial ? val + pv : val;
    else if (MODE == kResidual || MODE == kResidualRestrict)
      {
        const T av = post.a[il];
        val        = need_partial ? val + pv : val;
        return last ? av - val : val;
      }
    else
      {
        const T av = post.a[il], bv = post.b[il];
        T       xi = T(0), ov = T(0);
        if (MODE == kChebInit)
          xi = post.f0 * bv * av; // the same expression as in the gather: bitwise the same x_1
        else
          xi = src[il];
        if (MODE == kCheb)
          ov = post.old[il];
        else if (MODE == kChebOldInit)
          ov = post.f0 * bv * av;
        val  = need_partial ? val + pv : val;
        T xn = xi + post.f2 * bv * (av - val);
        if (MODE == kCheb || MODE == kChebOldInit)
          xn += post.f1 * (xi - ov);
        else if (MODE == kChebZeroOld || MODE == kChebInit)
          xn += post.f1 * xi;
        return last ? xn : val;
      }
  }

  // Cell-block order: the DoFs on the high side / in the interior of a cell (p^3 per cell, in
  // the order {hex, x-face, y-face, xy-line, z-face, xz-line, yz-line, vertex} = entities
  // 13,14,16,17,22,23,25,26 of the 27-entry table).  With a first-touch, cell-by-cell numbering
  // (mgx_cube; deal.II's matrix-free renumbering is of the same kind) these p^3 values are one
  // contiguous block per cell, so one wave instruction moves one 512-B cell block at p = 4.  The
  // mapping is correct for every numbering that satisfies the entity-contiguity contract; only
  // the coalescing depends on it.
  template <int P>
  __device__ __forceinline__ void decode_cell_dof(int kk, int &slot_rel, int &pnt_rel, int &k)
  {
    // kk in [0, p^3): find the entity (codes cx,cy,cz in {1,2}) and the offset inside it
    constexpr int G = BCfg<P>::G, E1 = BCfg<P>::NE1;
    int           off = 0;
    slot_rel = pnt_rel = k = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      {
        const int cx = 1 + (j & 1), cy = 1 + ((j >> 1) & 1), cz = 1 + (j >> 2);
        const int nx = cx == 1 ? P - 1 : 1, ny = cy == 1 ? P - 1 : 1, nz = cz == 1 ? P - 1 : 1;
        const int sz = nx * ny * nz;
        if (sz > 0 && kk >= off && kk < off + sz)
          {
            const int kl = kk - off;
            const int ox = kl % nx, oy = (kl / nx) % ny, oz = kl / (nx * ny);
            const int lx = cx == 1 ? 1 + ox : P, ly = cy == 1 ? 1 + oy : P, lz = cz == 1 ? 1 + oz : P;
            slot_rel = (cz * E1 + cy) * E1 + cx;
            pnt_rel  = (lz * G + ly) * G + lx;
            k        = kl;
          }
        off += sz;
      }
  }

  template <int P, typename T, int MODE, int NT>
  __device__ __forceinline__ void store_brick(int tid, T *acc, const uint32_t *ebase, const T *__restrict__ src,
                                              const BrickPost<T> &post)
  {
    using C          = BCfg<P>;
    constexpr int G  = C::G;
    constexpr int E1 = C::NE1;
    constexpr int P3 = P * P * P;
    constexpr int NA = C::NCELLS * P3;
    constexpr bool kFixedLane = (NT % P3) == 0; // per-thread decode hoisted out of the loops
    constexpr int NBF = G * G * G - (G - 1) * (G - 1) * (G - 1);
    constexpr int ITA = (NA + NT - 1) / NT;  // iterations of part A
    constexpr int ITB = (NBF + NT - 1) / NT; // iterations of part B
    // plain / residual forms only read partial sums of earlier launches: whole waves whose
    // entities are all FIRST skip pass 1
    constexpr bool kLoadsOnlyPartials = MODE == kPlain;
    int slot_rel = 0, pnt_rel = 0, k = 0;
    if (kFixedLane)
      decode_cell_dof<P>(tid % P3, slot_rel, pnt_rel, k);

    // Part A: the p^3 DoFs on the high side / in the interior of each cell, cell after cell.  With
    // a thread count that is a multiple of p^3 a thread keeps its position inside the cell block
    // and iteration `it` only advances the cell by a compile-time constant: the cell index is
    // m = m_thread + CPI * it with CPI = THREADS / p^3 a power of two, so the Morton bits of the
    // two parts are disjoint and (after full unrolling) the second part folds to immediates.
    constexpr int CPI = kFixedLane ? NT / P3 : 1;
    auto cell_slot = [](int m) {
      const int bx = (m & 1) | ((m >> 2) & 2), by = ((m >> 1) & 1) | ((m >> 3) & 2), bz = ((m >> 2) & 1) | ((m >> 4) & 2);
      return ((2 * bz) * E1 + 2 * by) * E1 + 2 * bx;
    };
    auto cell_pnt = [](int m) {
      const int bx = (m & 1) | ((m >> 2) & 2), by = ((m >> 1) & 1) | ((m >> 3) & 2), bz = ((m >> 2) & 1) | ((m >> 4) & 2);
      return ((bz * P) * G + by * P) * G + bx * P;
    };
    const int m_thread = kFixedLane ? tid / P3 : 0;
    const int e_thread = slot_rel + cell_slot(m_thread), pnt_thread = pnt_rel + cell_pnt(m_thread);
    auto item_a = [&](int it, int &e, int &pnt, uint32_t &off) {
      if (kFixedLane)
        {
          e   = e_thread + cell_slot(CPI * it);
          pnt = pnt_thread + cell_pnt(CPI * it);
          off = (uint32_t)k;
        }
      else
        {
          const int w = tid + it * NT, m = w / P3;
          decode_cell_dof<P>(w - m * P3, slot_rel, pnt_rel, k);
          e   = slot_rel + cell_slot(m);
          pnt = pnt_rel + cell_pnt(m);
          off = (uint32_t)k;
        }
    };
    // Part B: the low faces of the brick (points with a zero coordinate); their DoFs live in the
    // cell blocks of neighbouring bricks
    auto item_b = [&](int w, int &e, int &pnt, uint32_t &off) {
      int gx, gy, gz;
      if (w < G * G)
        {
          gz = 0;
          gy = w / G;
          gx = w - gy * G;
        }
      else if (w < G * G + G * (G - 1))
        {
          const int v = w - G * G;
          gy          = 0;
          gz          = 1 + v / G;
          gx          = v % G;
        }
      else
        {
          const int v = w - G * G - G * (G - 1);
          gx          = 0;
          gz          = 1 + v / (G - 1);
          gy          = 1 + v % (G - 1);
        }
      const int rx = gx % P, ry = gy % P, rz = gz % P;
      const int ex = 2 * (gx / P) + (rx != 0), ey = 2 * (gy / P) + (ry != 0), ez = 2 * (gz / P) + (rz != 0);
      const int nx = rx ? P - 1 : 1, ny = ry ? P - 1 : 1;
      const int ox = rx ? rx - 1 : 0, oy = ry ? ry - 1 : 0, oz = rz ? rz - 1 : 0;
      e   = (ez * E1 + ey) * E1 + ex;
      pnt = (gz * G + gy) * G + gx;
      off = (uint32_t)((oz * ny + oy) * nx + ox);
    };

    // ---- pass 1: loads + post-operation, result back into the accumulator (same thread).  The
    // table words (and for part B the decoded positions) stay in registers for pass 2. ----
    constexpr bool kKeepA = ITA <= 32;
    uint32_t       wa[kKeepA ? ITA : 1];
#pragma unroll
    for (int it = 0; it < ITA; ++it)
      {
        if (tid + it * NT >= NA)
          break;
        int      e, pnt;
        uint32_t off;
        item_a(it, e, pnt, off);
        const uint32_t w = ebase[e];
        if (kKeepA)
          wa[it] = w;
        const bool     valid = w != kInvalid;
        const uint32_t fl    = ent_flags_of(w);
        if (kLoadsOnlyPartials && !__any(valid && !(fl & 1)))
          continue;
        acc[pnt] = post_value<P, T, MODE>(src, post, valid, ent_index(w) + off, (uint8_t)fl, acc[pnt]);
      }
    uint32_t wb[ITB], ib[ITB];
    int      pb[ITB];
#pragma unroll
    for (int it = 0; it < ITB; ++it)
      {
        wb[it] = kInvalid;
        ib[it] = 0;
        pb[it] = 0;
        const int w_item = tid + it * NT;
        if (w_item < NBF)
          {
            int      e;
            uint32_t off;
            item_b(w_item, e, pb[it], off);
            const uint32_t w = ebase[e];
            wb[it]           = w;
            ib[it]           = ent_index(w) + off;
            acc[pb[it]] = post_value<P, T, MODE>(src, post, w != kInvalid, ib[it], (uint8_t)ent_flags_of(w), acc[pb[it]]);
          }
      }
    // ---- pass 2: stores only.  With a first-touch cell-by-cell numbering part A writes one
    // contiguous block per cell (one 512-B wave instruction at p = 4). ----
#pragma unroll
    for (int it = 0; it < ITA; ++it)
      {
        if (tid + it * NT >= NA)
          break;
        int      e, pnt;
        uint32_t off;
        item_a(it, e, pnt, off);
        const uint32_t w = kKeepA ? wa[it] : ebase[e];
        if (MODE == kResidualRestrict)
          {
            // completed residuals stay in the accumulator for the restriction; partial sums go to
            // the carrier; everything that is not a completed residual becomes zero
            if (w == kInvalid)
              acc[pnt] = T(0);
            else if (!(w >> 31))
              {
                post.partial[ent_index(w) + off] = acc[pnt];
                acc[pnt]                         = T(0);
              }
          }
        else if (w != kInvalid)
          ((w >> 31) ? post.out : post.partial)[ent_index(w) + off] = acc[pnt];
      }
#pragma unroll
    for (int it = 0; it < ITB; ++it)
      {
        if (MODE == kResidualRestrict)
          {
            if (tid + it * NT < NBF)
              {
                if (wb[it] == kInvalid)
                  acc[pb[it]] = T(0);
                else if (!(wb[it] >> 31))
                  {
                    post.partial[ib[it]] = acc[pb[it]];
                    acc[pb[it]]          = T(0);
                  }
              }
          }
        else if (wb[it] != kInvalid)
          ((wb[it] >> 31) ? post.out : post.partial)[ib[it]] = acc[pb[it]];
      }
  }




  template <int P, typename T, int MODE>
  __global__ void __launch_bounds__(BCfg<P>::THREADS)
    brick_loop_kernel(const T *__restrict__ src, uint32_t brick_first, const uint32_t *__restrict__ ent_base,
                      const uint8_t *__restrict__ ent_flags, const Basis1D<T> *__restrict__ B, T c0, T c1, T c2,
                      BrickPost<T> post)
  {
    using C          = BCfg<P>;
    constexpr int N  = C::N;
    constexpr int LN = C::LN;
    constexpr int PL = N * LN;
    constexpr int G  = C::G;
    constexpr int E1 = C::NE1;
    static_assert(C::TPC <= 32 && C::THREADS == 256, "two cells per wave, four waves");
    __shared__ T        acc[G * G * G];
    __shared__ T        U[8 * C::CELL_LDS];
    __shared__ uint32_t ebase[C::NE];

    const int      tid   = threadIdx.x;
    const uint32_t brick = brick_first + blockIdx.x;
    for (int i = tid; i < C::NE; i += C::THREADS)
      ebase[i] = ent_base[(size_t)brick * C::NE + i];
    for (int i = tid; i < G * G * G; i += C::THREADS)
      acc[i] = T(0);

    // a wave owns two cells of every round: lanes [0,TPC) and [32,32+TPC)
    const int  lane    = tid & 63;
    const int  t       = lane & 31;
    const bool compute = t < C::TPC;
    const int  lc      = 2 * (tid >> 6) + (lane >> 5); // 0..7
    const int  a       = compute ? t % N : 0;
    const int  b       = compute ? t / N : 0;
    T         *Uc      = U + lc * C::CELL_LDS;
    const int  xl      = (b * N + a) * LN;
    const int  yl      = b * PL + a;
    const int  zl      = b * LN + a;
    const T    wa = B->w[a], wb = B->w[b];
    // entity codes of this thread's x-line (j = a, k = b): vector_access_reduced.h:232-247
    const int cy = (a == 0) ? 0 : (a == P ? 2 : 1), oy = (cy == 1) ? a - 1 : 0;
    const int cz = (b == 0) ? 0 : (b == P ? 2 : 1), oz = (cz == 1) ? b - 1 : 0;
    const uint32_t loff = (uint32_t)((cy == 1 ? P - 1 : 1) * oz + oy);
    // position of the cell inside its 2x2x2 octant (lc bits); the round adds the parity
    const int hx = 2 * (lc & 1), hy = 2 * ((lc >> 1) & 1), hz = 2 * (lc >> 2);
    __syncthreads();

    // read_dof_values_compressed through the brick's entity table for the cell of `round`.
    // The loads are unconditional (constrained entities read element 0 and are masked when
    // consumed) so that no branch forces a wait: they stay in flight as a prefetch.
    uint32_t nvalid = 0; // bit i: entity i of the prefetched line is unconstrained
    auto gather = [&](int round, T(&r)[N]) {
      const int       bx = hx + (round & 1), by = hy + ((round >> 1) & 1), bz = hz + (round >> 2);
      const uint32_t *eb = ebase + ((2 * bz + cz) * E1 + (2 * by + cy)) * E1 + 2 * bx;
      const uint32_t  b0 = eb[0], b1 = eb[1], b2 = eb[2];
      nvalid = (b0 != kInvalid ? 1u : 0u) | (b1 != kInvalid ? 2u : 0u) | (b2 != kInvalid ? 4u : 0u);
      r[0] = src[b0 != kInvalid ? ent_index(b0) + loff : 0u];
      const uint32_t m1 = b1 != kInvalid ? ent_index(b1) + loff * (uint32_t)(P - 1) : 0u;
#pragma unroll
      for (int i = 0; i < P - 1; ++i)
        r[1 + i] = src[m1 + (uint32_t)i];
      r[P] = src[b2 != kInvalid ? ent_index(b2) + loff : 0u];
    };

    // Source values of the next round (prefetched).  The gather runs in uniform control flow
    // (idle lanes duplicate line (0,0)): defined inside the lane-masked region the compiler
    // would have to wait for the loads at the end of that region instead of at the next use.
    T rn[N];
    gather(0, rn);

#pragma unroll 1
    for (int round = 0; round < 8; ++round)
      {
        T r[N], q[N], vx[N], vy[N], vz[N];
        r[0] = (nvalid & 1u) ? rn[0] : T(0);
#pragma unroll
        for (int i = 1; i < P; ++i)
          r[i] = (nvalid & 2u) ? rn[i] : T(0);
        r[P] = (nvalid & 4u) ? rn[P] : T(0);
        if (round < 7)
          gather(round + 1, rn); // in flight during the whole round
        if (compute)
          {
            // 1. nodal -> quadrature along x
            bmv<N, T>(B->S, r, q);
#pragma unroll
            for (int i = 0; i < N; ++i)
              Uc[xl + i] = q[i];
          }
        wave_sync();
        if (compute) // 2. along y
          {
#pragma unroll
            for (int i = 0; i < N; ++i)
              r[i] = Uc[yl + i * LN];
            bmv<N, T>(B->S, r, q);
#pragma unroll
            for (int i = 0; i < N; ++i)
              Uc[yl + i * LN] = q[i];
          }
        wave_sync();
        if (compute) // 3. along z; z-derivative pair stays in registers
          {
#pragma unroll
            for (int i = 0; i < N; ++i)
              r[i] = Uc[zl + i * PL];
            bmv<N, T>(B->S, r, q);
#pragma unroll
            for (int i = 0; i < N; ++i)
              Uc[zl + i * PL] = q[i];
            bmv<N, T>(B->D, q, r);
            const T f = c2 * wa * wb;
#pragma unroll
            for (int i = 0; i < N; ++i)
              r[i] *= f * B->w[i];
            bmvT<N, T>(B->D, r, vz);
          }
        wave_sync();
        if (compute) // 4./5. x- and y-derivative pairs from the quadrature values (read only)
          {
#pragma unroll
            for (int i = 0; i < N; ++i)
              q[i] = Uc[xl + i];
            bmv<N, T>(B->D, q, r);
            const T fx = c0 * wa * wb;
#pragma unroll
            for (int i = 0; i < N; ++i)
              r[i] *= fx * B->w[i];
            bmvT<N, T>(B->D, r, vx);
#pragma unroll
            for (int i = 0; i < N; ++i)
              q[i] = Uc[yl + i * LN];
            bmv<N, T>(B->D, q, r);
            const T fy = c1 * wa * wb;
#pragma unroll
            for (int i = 0; i < N; ++i)
              r[i] *= fy * B->w[i];
            bmvT<N, T>(B->D, r, vy);
          }
        wave_sync();
        if (compute) // 6. the buffer is free: deposit the x part
          {
#pragma unroll
            for (int i = 0; i < N; ++i)
              Uc[xl + i] = vx[i];
          }
        wave_sync();
        if (compute) // 7. add the y part
          {
#pragma unroll
            for (int i = 0; i < N; ++i)
              Uc[yl + i * LN] += vy[i];
          }
        wave_sync();
        if (compute) // 8. add the z part, integrate along z
          {
#pragma unroll
            for (int i = 0; i < N; ++i)
              r[i] = Uc[zl + i * PL] + vz[i];
            bmvT<N, T>(B->S, r, q);
#pragma unroll
            for (int i = 0; i < N; ++i)
              Uc[zl + i * PL] = q[i];
          }
        wave_sync();
        if (compute) // 9. along y
          {
#pragma unroll
            for (int i = 0; i < N; ++i)
              r[i] = Uc[yl + i * LN];
            bmvT<N, T>(B->S, r, q);
#pragma unroll
            for (int i = 0; i < N; ++i)
              Uc[yl + i * LN] = q[i];
          }
        wave_sync();
        if (compute) // 10. along x
          {
#pragma unroll
            for (int i = 0; i < N; ++i)
              r[i] = Uc[xl + i];
            bmvT<N, T>(B->S, r, q);
          }
        // cells of consecutive rounds are neighbours: all accumulator updates of the previous
        // round must have landed before this round's begin
        lds_barrier();
        if (compute)
          {
            // distribute_local_to_global into the brick accumulator: no other cell of this
            // round touches these points
            const int bx = hx + (round & 1), by = hy + ((round >> 1) & 1), bz = hz + (round >> 2);
            T        *row = acc + ((bz * P + b) * G + (by * P + a)) * G + bx * P;
#pragma unroll
            for (int i = 0; i < N; ++i)
              row[i] += q[i];
          }
      }
    __syncthreads();

    store_brick<P, T, MODE, C::THREADS>(tid, acc, ebase, src, post);
  }

  // ------------------------------------------------------------------------------------------
  // Separable fast path.  On a Cartesian mesh with a constant coefficient the merged coefficient
  // is one diagonal tensor for the whole mesh (laplace_operator.h:374-387, 447-491) and the
  // quadrature weights factorise, so   S^T [sum_d D_d^T (c_d w) D_d] S  =  sum_d c_d (M x M x K_d)
  // with the 1D matrices M = S^T W S and K = S^T D^T W D S:
  //     t1 = M_x u, k1 = K_x u ;  t2 = M_y t1, s2 = c_x M_y k1 + c_y K_y t1 ;
  //     out = M_z s2 + c_z K_z t2
  // 7 sweeps instead of 12 and 2 LDS transposes instead of 8; M and K are symmetric and
  // persymmetric, so every sweep runs in even-odd form with coefficients that stay in SGPRs.
  // ------------------------------------------------------------------------------------------

  template <int P, typename T, int MODE, bool WIDE>
  __global__ void __launch_bounds__((BCfg<P, WIDE>::THREADS))
    brick_sep_kernel(const T *__restrict__ src, uint32_t brick_first, const uint32_t *__restrict__ ent_base,
                     const uint8_t *__restrict__ ent_flags, const Basis1D<T> *__restrict__ B, T c0, T c1, T c2,
                     BrickPost<T> post)
  {
    using C          = BCfg<P, WIDE>;
    constexpr int N  = C::N;
    constexpr int LN = C::LN;
    constexpr int PL = N * LN;
    constexpr int G  = C::G;
    constexpr int E1 = C::NE1;
    constexpr int H1 = N / 2 + 1;
    __shared__ T        acc[G * G * G];
    // fp32: two transpose buffers per cell, so that both arrays of a transpose move in one
    // write / one read phase (half the LDS round trips per round); the fp64 accumulator leaves no
    // room for that at 3 workgroups per CU
    constexpr bool kDualU = sizeof(T) == 4;
    __shared__ T        U[(kDualU ? 2 : 1) * C::ROUND_CELLS * C::CELL_LDS];
    __shared__ uint32_t ebase[C::NE];

    const int      tid   = threadIdx.x;
    const uint32_t brick = brick_first + blockIdx.x;
    for (int i = tid; i < C::NE; i += C::THREADS)
      ebase[i] = ent_base[(size_t)brick * C::NE + i];
    for (int i = tid; i < G * G * G; i += C::THREADS)
      acc[i] = T(0);

    // p <= 4: lanes [0,TPC) and [32,32+TPC) of each wave own the two cells of that wave;
    // p >= 5: each half of the workgroup owns one cell per round
    const int  lane    = tid & 63;
    const int  t       = C::kTwoPerWave ? (lane & 31) : tid % C::TPW;
    const bool compute = t < C::TPC;
    const int  lc      = C::kTwoPerWave ? 2 * (tid >> 6) + (lane >> 5) : tid / C::TPW;
    const int  a       = compute ? t % N : 0;
    const int  b       = compute ? t / N : 0;
    T         *Uc      = U + lc * C::CELL_LDS;
    T         *Ud      = U + (kDualU ? C::ROUND_CELLS + lc : lc) * C::CELL_LDS; // second buffer (fp32)
    const int  xl      = (b * N + a) * LN;
    const int  yl      = b * PL + a;
    const int  zl      = b * LN + a;
    const int cy = (a == 0) ? 0 : (a == P ? 2 : 1), oy = (cy == 1) ? a - 1 : 0;
    const int cz = (b == 0) ? 0 : (b == P ? 2 : 1), oz = (cz == 1) ? b - 1 : 0;
    const uint32_t loff = (uint32_t)((cy == 1 ? P - 1 : 1) * oz + oy);
    const int hx = 2 * (lc & 1), hy = 2 * ((lc >> 1) & 1), hz = 2 * ((lc >> 2) & 1);
    const int half = C::kTwoPerWave ? (lc >> 3) : lc; // which of the two cell sets of a round (if two)
    const EOMat<T> &M = B->mass, &K = B->lapl;
    // cell of this thread in a round: p <= 4 the same-parity cell of its 2x2x2 octant, p >= 5 cell
    // 2 round + lc of the 2x2x2 brick
    auto cell_of = [&](int round, int &bx, int &by, int &bz) {
      if (C::kTwoPerWave)
        {
          const int r = C::kWide ? 2 * round + half : round; // parity class
          bx          = hx + (r & 1);
          by          = hy + ((r >> 1) & 1);
          bz          = hz + (r >> 2);
        }
      else
        {
          const int m = 2 * round + lc;
          bx          = m & 1;
          by          = (m >> 1) & 1;
          bz          = m >> 2;
        }
    };
    // ordering of the LDS transposes: inside one wave a compiler barrier is enough, a cell spread
    // over two waves needs the workgroup barrier
    auto phase_sync = [&]() {
      if (C::kWaveSync)
        wave_sync();
      else
        lds_barrier();
    };
    __syncthreads();

    // source values are prefetched two rounds ahead (two register sets, statically indexed by
    // unrolling the round loop by two): an HBM miss takes about as long as one round
    constexpr int ND = MODE == kChebInit ? N : 1; // kChebInit gathers two operands per value
    auto gather = [&](int round, T(&r)[N], T(&d)[ND], uint32_t &nvalid) {
      int bx, by, bz;
      cell_of(round, bx, by, bz);
      const uint32_t *eb = ebase + ((2 * bz + cz) * E1 + (2 * by + cy)) * E1 + 2 * bx;
      const uint32_t  b0 = eb[0], b1 = eb[1], b2 = eb[2];
      nvalid = (b0 != kInvalid ? 1u : 0u) | (b1 != kInvalid ? 2u : 0u) | (b2 != kInvalid ? 4u : 0u);
      const T        *v0 = MODE == kChebInit ? post.a : src;
      const uint32_t  m0 = b0 != kInvalid ? ent_index(b0) + loff : 0u;
      const uint32_t  m1 = b1 != kInvalid ? ent_index(b1) + loff * (uint32_t)(P - 1) : 0u;
      const uint32_t  m2 = b2 != kInvalid ? ent_index(b2) + loff : 0u;
      r[0] = v0[m0];
#pragma unroll
      for (int i = 0; i < P - 1; ++i)
        r[1 + i] = v0[m1 + (uint32_t)i];
      r[P] = v0[m2];
      if (MODE == kChebInit)
        {
          d[0] = post.b[m0];
#pragma unroll
          for (int i = 0; i < P - 1; ++i)
            d[(1 + i) % ND] = post.b[m1 + (uint32_t)i];
          d[P % ND] = post.b[m2];
        }
    };
    T        rA[N], rB[N], dA[ND], dB[ND];
    uint32_t vA = 0, vB = 0;
    gather(0, rA, dA, vA);
    gather(1, rB, dB, vB);

    auto one_round = [&](int round, T(&rn)[N], T(&dn)[ND], uint32_t &nvalid) {
        T r[N], t1[N], k1[N], xe[H1], xo[H1];
        if (MODE == kChebInit)
          {
#pragma unroll
            for (int i = 0; i < N; ++i)
              rn[i] = post.f0 * dn[i % ND] * rn[i]; // x_1 = f0 D^-1 b, as in post_value
          }
        r[0] = (nvalid & 1u) ? rn[0] : T(0);
#pragma unroll
        for (int i = 1; i < P; ++i)
          r[i] = (nvalid & 2u) ? rn[i] : T(0);
        r[P] = (nvalid & 4u) ? rn[P] : T(0);
        if (round + 2 < C::ROUNDS)
          gather(round + 2, rn, dn, nvalid); // this register set is free again: refill it
        // x: t1 = M u, k1 = K u
        eo_split<N, T>(r, xe, xo);
        eo_apply<N, T>(M, xe, xo, t1);
        eo_apply<N, T>(K, xe, xo, k1);
        if (kDualU)
          {
            if (compute)
              {
#pragma unroll
                for (int i = 0; i < N; ++i)
                  {
                    Uc[xl + i] = t1[i];
                    Ud[xl + i] = k1[i];
                  }
              }
            phase_sync();
#pragma unroll
            for (int i = 0; i < N; ++i)
              {
                t1[i] = Uc[yl + i * LN]; // y-lines of M_x u and K_x u
                k1[i] = Ud[yl + i * LN];
              }
            phase_sync();
          }
        else
          {
            if (compute)
              {
#pragma unroll
                for (int i = 0; i < N; ++i)
                  Uc[xl + i] = t1[i];
              }
            phase_sync();
#pragma unroll
            for (int i = 0; i < N; ++i)
              t1[i] = Uc[yl + i * LN]; // y-line of M_x u
            phase_sync();
            if (compute)
              {
#pragma unroll
                for (int i = 0; i < N; ++i)
                  Uc[xl + i] = k1[i];
              }
            phase_sync();
#pragma unroll
            for (int i = 0; i < N; ++i)
              k1[i] = Uc[yl + i * LN]; // y-line of K_x u
            phase_sync();
          }
        // y: t2 = M t1 ; s2 = c_x M k1 + c_y K t1
        T t2[N], s2[N];
        eo_split<N, T>(t1, xe, xo);
        eo_apply<N, T>(M, xe, xo, t2);
        eo_apply<N, T>(K, xe, xo, s2);
        eo_split<N, T>(k1, xe, xo);
        eo_apply<N, T>(M, xe, xo, r);
#pragma unroll
        for (int i = 0; i < N; ++i)
          s2[i] = fma(c0, r[i], c1 * s2[i]);
        if (kDualU)
          {
            if (compute)
              {
#pragma unroll
                for (int i = 0; i < N; ++i)
                  {
                    Uc[yl + i * LN] = t2[i];
                    Ud[yl + i * LN] = s2[i];
                  }
              }
            phase_sync();
#pragma unroll
            for (int i = 0; i < N; ++i)
              {
                t2[i] = Uc[zl + i * PL]; // z-lines
                s2[i] = Ud[zl + i * PL];
              }
            phase_sync(); // the next round overwrites both buffers
          }
        else
          {
            if (compute)
              {
#pragma unroll
                for (int i = 0; i < N; ++i)
                  Uc[yl + i * LN] = t2[i];
              }
            phase_sync();
#pragma unroll
            for (int i = 0; i < N; ++i)
              t2[i] = Uc[zl + i * PL]; // z-line of M_y M_x u
            phase_sync();
            if (compute)
              {
#pragma unroll
                for (int i = 0; i < N; ++i)
                  Uc[yl + i * LN] = s2[i];
              }
            phase_sync();
#pragma unroll
            for (int i = 0; i < N; ++i)
              s2[i] = Uc[zl + i * PL];
          }
        // z: out = M s2 + c_z K t2
        eo_split<N, T>(s2, xe, xo);
        eo_apply<N, T>(M, xe, xo, r);
        eo_split<N, T>(t2, xe, xo);
        eo_apply<N, T>(K, xe, xo, t1);
#pragma unroll
        for (int i = 0; i < N; ++i)
          r[i] = fma(c2, t1[i], r[i]);
        lds_barrier(); // all accumulator updates of the previous round have landed
          {
            // thread (i = a, j = b) owns the z-line: accumulate the column of the brick array.
            // p >= 5: the two cells of the round touch, the second half adds after the first
            int bx, by, bz;
            cell_of(round, bx, by, bz);
            T *col = acc + ((bz * P) * G + (by * P + b)) * G + bx * P + a;
            constexpr bool kTwoSets = !C::kTwoPerWave || C::kWide;
            if (compute && (!kTwoSets || half == 0))
              {
#pragma unroll
                for (int i = 0; i < N; ++i)
                  col[i * G * G] += r[i];
              }
            if (kTwoSets)
              {
                lds_barrier();
                if (compute && half == 1)
                  {
#pragma unroll
                    for (int i = 0; i < N; ++i)
                      col[i * G * G] += r[i];
                  }
              }
          }
    };
      {
#pragma unroll 1
        for (int round = 0; round < C::ROUNDS; round += 2)
          {
            one_round(round, rA, dA, vA);
            one_round(round + 1, rB, dB, vB);
          }
      }
    __syncthreads();

    store_brick<P, T, MODE, C::THREADS>(tid, acc, ebase, src, post);
    if (MODE == kResidualRestrict)
      {
        constexpr int CE1 = C::NB + 1; // 2 PB + 1
        __syncthreads();
        restrict_brick<P, T, C::THREADS>(tid, acc, B->P1eo, post.coarse, post.coarse_blocks + (size_t)brick * (CE1 * CE1 * CE1));
      }
  }


  template <int P, typename T, int MODE>
  static void brick_launch(hipStream_t s, const OperatorData &op, const T *src, const BrickPost<T> &post, int g0, int g1)
  {
    using C               = BCfg<P>;
    const BrickData &bd   = op.bricks;
    for (int c = g0; c < g1; ++c)
      {
        const uint32_t first = bd.colour_start[c], count = bd.colour_start[c + 1] - first;
        if (count == 0)
          continue;
        if (op.separable)
          {
            // few bricks per launch (< 4 per CU): the launch lasts one workgroup's latency, which the
            // 512-thread form roughly halves; with the chip full the 256-thread form is faster
            if (P <= 4 && count < op.wide_max)
              hipLaunchKernelGGL((brick_sep_kernel<P, T, MODE, true>), dim3(count), dim3(BCfg<P, true>::THREADS), 0,
                                 s, src, first, bd.ent_base, bd.ent_flags, (const Basis1D<T> *)op.basis,
                                 (T)op.coef[0], (T)op.coef[1], (T)op.coef[2], post);
            else
              hipLaunchKernelGGL((brick_sep_kernel<P, T, MODE, false>), dim3(count), dim3(C::THREADS), 0, s, src,
                                 first, bd.ent_base, bd.ent_flags, (const Basis1D<T> *)op.basis, (T)op.coef[0],
                                 (T)op.coef[1], (T)op.coef[2], post);
          }
        else if constexpr (P <= 4 && MODE != kChebInit && MODE != kChebOldInit && MODE != kResidualRestrict) // quadrature-point form: 4x4x4 bricks only
          hipLaunchKernelGGL((brick_loop_kernel<P, T, MODE>), dim3(count),
                             dim3(C::THREADS), 0, s, src, first, bd.ent_base, bd.ent_flags,
                             (const Basis1D<T> *)op.basis, (T)op.coef[0], (T)op.coef[1], (T)op.coef[2], post);
      }
  }

  template <typename T>
  static void brick_dispatch(hipStream_t s, const OperatorData &op, int mode, const void *src, const void *a,
                             const void *b, void *out, void *partial, double f1, double f2, const void *old, double f0,
                             void *coarse, const uint32_t *coarse_blocks, int g0, int g1)
  {
    BrickPost<T> post;
    post.f0      = (T)f0;
    post.coarse  = (T *)coarse;
    post.coarse_blocks = coarse_blocks;
    post.a       = (const T *)a;
    post.b       = (const T *)b;
    post.old     = (const T *)old;
    post.out     = (T *)out;
    post.partial = (T *)partial;
    post.f1      = (T)f1;
    post.f2      = (T)f2;
#define MGX_BRICK_CASE(PP)                                                         \
  case PP:                                                                         \
    switch (mode)                                                                  \
      {                                                                            \
        case kPlain: brick_launch<PP, T, kPlain>(s, op, (const T *)src, post, g0, g1); break; \
        case kResidual: brick_launch<PP, T, kResidual>(s, op, (const T *)src, post, g0, g1); break; \
        case kCheb: brick_launch<PP, T, kCheb>(s, op, (const T *)src, post, g0, g1); break; \
        case kChebFirst: brick_launch<PP, T, kChebFirst>(s, op, (const T *)src, post, g0, g1); break; \
        case kChebInit: brick_launch<PP, T, kChebInit>(s, op, (const T *)src, post, g0, g1); break; \
        case kChebOldInit: brick_launch<PP, T, kChebOldInit>(s, op, (const T *)src, post, g0, g1); break; \
        case kResidualRestrict: brick_launch<PP, T, kResidualRestrict>(s, op, (const T *)src, post, g0, g1); break; \
        default: brick_launch<PP, T, kChebZeroOld>(s, op, (const T *)src, post, g0, g1); break; \
      }                                                                            \
    break;
    switch (op.p)
      {
        MGX_BRICK_CASE(1)
        MGX_BRICK_CASE(2)
        MGX_BRICK_CASE(3)
        MGX_BRICK_CASE(4)
        MGX_BRICK_CASE(5)
        MGX_BRICK_CASE(6)
        MGX_BRICK_CASE(7)
        MGX_BRICK_CASE(8)
        MGX_BRICK_CASE(9)
        default: break;
      }
#undef MGX_BRICK_CASE
  }
#endif // MGX_CELLS_FORM

  void launch_brick_loop(hipStream_t s, const OperatorData &op, int mode, const void *src, const void *a,
                         const void *b, void *out, void *partial, double f1, double f2, const void *old, double f0,
                         void *coarse, const uint32_t *coarse_blocks, int g0, int g1, bool free_schedule)
  {
    if (g1 < 0)
      g1 = op.bricks.n_colours;
    if (!old)
      old = out;
    if (!src)
      src = (const void *)a; // kChebInit: never dereferenced, but keep the pointer valid
    // separable operator: macro-element form (mgx_macro.hip); Tunables::cells_form keeps the
    // cell-by-cell form below (A/B measurements, and the reference point of the consistency tests)
    if (op.separable && op.bricks.item_map && !op.cells_form)
      {
        const bool done = op.number == 1
                            ? launch_macro_loop_f64(s, op, mode, src, a, b, out, partial, f1, f2, old, f0, coarse, coarse_blocks, g0, g1, free_schedule)
                            : launch_macro_loop_f32(s, op, mode, src, a, b, out, partial, f1, f2, old, f0, coarse, coarse_blocks, g0, g1, free_schedule);
        if (done)
          return;
      }
    // (the caller asks for the reduced-colour schedules only where the macro-element kernel runs)
#if MGX_CELLS_FORM
    if (op.number == 1)
      brick_dispatch<double>(s, op, mode, src, a, b, out, partial, f1, f2, old, f0, coarse, coarse_blocks, g0, g1);
    else
      brick_dispatch<float>(s, op, mode, src, a, b, out, partial, f1, f2, old, f0, coarse, coarse_blocks, g0, g1);
#else
    // unreachable: mgx_operator_create builds a brick schedule only for operators the macro-element
    // kernel covers; fail loudly rather than return without having computed anything
    fprintf(stderr, "mgx: brick loop requested for an operator the macro-element kernel does not cover (mode %d)\n", mode);
    abort();
#endif
  }
} // namespace mgx
