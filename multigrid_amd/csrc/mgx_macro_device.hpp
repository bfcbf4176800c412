// mgx_macro_device.hpp -- device code shared by the two pipelines of the macro-element brick loop
// (mgx_macro.hip: gather after the sweeps, all modes; mgx_macro2.hip: gather one brick ahead, before the sweeps):
// launch configuration, the assembled 1D line products in even-odd form, buffer-descriptor access, the fused
// post-operation.
#pragma once

#include "mgx_brick_device.hpp"
#include "mgx_bricks.hpp" // MGX_MACRO_PAIRS

#include <hip/hip_runtime.h>

namespace mgx
{
  template <int P, typename T>
  struct MCfg
  {
    using C = BCfg<P>;
    static constexpr int NB = C::NB, G = C::G, NE = C::NE, N = P + 1;
    static constexpr int LINES   = G * G;
    // One thread per line, except that a small remainder of lines beyond a multiple of 64 goes to
    // the first wave as a second pass instead of to a wave of its own (G = 17: 289 lines on 256
    // threads).  Measured on MI355X (tools/experiments/occupancy_probe.hip): workgroups of 5 or 6
    // waves are admitted one per CU only as soon as they use more than 48 kB of LDS, workgroups of
    // 2, 3, 4 or 8 waves two per CU up to 80 kB each.
    static constexpr int THREADS = (LINES > 128 && LINES % 64 <= 40) ? (LINES / 64) * 64 : ((LINES + 63) / 64) * 64;
    static constexpr int NPTS    = G * G * G;
    // item table (mgx_bricks.hpp): NPAIR pairs of adjacent DoFs, then NSINGLE single DoFs.  Entities
    // with t cell-interior directions: C(3,t) NB^t (NB+1)^(3-t) of them, (p-1)^t DoFs each
    static constexpr int ent_count(int t) { return (t == 0 || t == 3 ? 1 : 3) * ipow(NB, t) * ipow(NB + 1, 3 - t); }
    static constexpr int ipow(int b, int e) { return e == 0 ? 1 : b * ipow(b, e - 1); }
    static constexpr int ent_size(int t) { return ipow(P - 1, t); }
    static constexpr int NPAIR   = MGX_MACRO_PAIRS ? ent_count(0) * (ent_size(0) / 2) + ent_count(1) * (ent_size(1) / 2) +
                                                     ent_count(2) * (ent_size(2) / 2) + ent_count(3) * (ent_size(3) / 2)
                                                 : 0;
    static constexpr int NSINGLE = NPTS - 2 * NPAIR;
    static constexpr int JP      = (NPAIR + THREADS - 1) / THREADS;   // pairs per thread
    static constexpr int JS      = (NSINGLE + THREADS - 1) / THREADS; // singles per thread
    static constexpr int IT      = 2 * JP + JS;                       // value slots per thread
    static constexpr int TSZ     = (3 * NE * 4 + (int)sizeof(T) - 1) / (int)sizeof(T); // two entity tables + the surface table, in T's
    static constexpr int ASZ     = NPTS > TSZ ? NPTS : TSZ;                            // LDS array size
    static constexpr int LDS     = 2 * ASZ * (int)sizeof(T);
    static constexpr int WG_LDS  = 163840 / LDS;
    static constexpr int WG_WAVE = 32 / (THREADS / 64);
    static constexpr int WGS     = WG_LDS < WG_WAVE ? (WG_LDS < 8 ? WG_LDS : 8) : (WG_WAVE < 8 ? WG_WAVE : 8);
    static constexpr int WAVES   = WGS * (THREADS / 64);
    // Waves per SIMD the register allocation is asked to allow: what WGS workgroups per CU need, but
    // not more than the registers of the line sweeps (three lines of G values) and of the per-item
    // state (source value, inverse diagonal, item word) leave room for -- spilling costs more than a
    // wave per SIMD
    static constexpr int REGS = (3 * G + 2 * IT) * ((int)sizeof(T) / 4) + 2 * IT + 40;
    static constexpr int RMAX = REGS > 168 ? 2 : (REGS > 128 ? 3 : 4);
    static constexpr int MINW = (WAVES + 3) / 4 < RMAX ? (WAVES + 3) / 4 : RMAX;
  };

  // out = Ab in for the assembled 1D matrix of A over the NB cells of a line, cell block by cell
  // block in even-odd form (coefficients wave-uniform: scalar registers)
  template <int P, typename T>
  __device__ __forceinline__ void macro_apply(const EOMat<T> &A, const T (&in)[BCfg<P>::G], T (&out)[BCfg<P>::G])
  {
    constexpr int N = P + 1, NB = BCfg<P>::NB, H1 = N / 2 + 1;
#pragma unroll
    for (int c = 0; c < NB; ++c)
      {
        T seg[N], xe[H1], xo[H1], y[N];
#pragma unroll
        for (int i = 0; i < N; ++i)
          seg[i] = in[c * P + i];
        eo_split<N, T>(seg, xe, xo);
        eo_apply<N, T>(A, xe, xo, y);
        if (c == 0)
          out[0] = y[0];
        else
          out[c * P] += y[0];
#pragma unroll
        for (int i = 1; i < N; ++i)
          out[c * P + i] = y[i];
      }
  }

  // the same for two matrices applied to one line (the even-odd split is shared)
  template <int P, typename T>
  __device__ __forceinline__ void macro_apply2(const EOMat<T> &A, const EOMat<T> &Bm, const T (&in)[BCfg<P>::G],
                                               T (&oa)[BCfg<P>::G], T (&ob)[BCfg<P>::G])
  {
    constexpr int N = P + 1, NB = BCfg<P>::NB, H1 = N / 2 + 1;
#pragma unroll
    for (int c = 0; c < NB; ++c)
      {
        T seg[N], xe[H1], xo[H1], y[N], z[N];
#pragma unroll
        for (int i = 0; i < N; ++i)
          seg[i] = in[c * P + i];
        eo_split<N, T>(seg, xe, xo);
        eo_apply<N, T>(A, xe, xo, y);
        eo_apply<N, T>(Bm, xe, xo, z);
        if (c == 0)
          {
            oa[0] = y[0];
            ob[0] = z[0];
          }
        else
          {
            oa[c * P] += y[0];
            ob[c * P] += z[0];
          }
#pragma unroll
        for (int i = 1; i < N; ++i)
          {
            oa[c * P + i] = y[i];
            ob[c * P + i] = z[i];
          }
      }
  }

  // one cell block of a line: y = A seg (even-odd form)
  template <int P, typename T>
  __device__ __forceinline__ void cell_apply(const EOMat<T> &A, const T (&seg)[P + 1], T (&y)[P + 1])
  {
    constexpr int N = P + 1, H1 = N / 2 + 1;
    T             xe[H1], xo[H1];
    eo_split<N, T>(seg, xe, xo);
    eo_apply<N, T>(A, xe, xo, y);
  }
  template <int P, typename T>
  __device__ __forceinline__ void cell_apply2(const EOMat<T> &A, const EOMat<T> &Bm, const T (&seg)[P + 1],
                                              T (&y)[P + 1], T (&z)[P + 1])
  {
    constexpr int N = P + 1, H1 = N / 2 + 1;
    T             xe[H1], xo[H1];
    eo_split<N, T>(seg, xe, xo);
    eo_apply<N, T>(A, xe, xo, y);
    eo_apply<N, T>(Bm, xe, xo, z);
  }
  // value of the next lane (the next cell of the same line in the cell-split pass)
  __device__ __forceinline__ double next_lane(double v) { return __shfl_down(v, 1); }
  __device__ __forceinline__ float  next_lane(float v) { return __shfl_down(v, 1); }

  // ------------------------------------------------------------------------------------------
  // Sliced sweeps (high degrees).  The two even-odd matrices of a sweep are 2 x 41 doubles at p = 8: 164 scalar
  // registers where ~100 exist, and what does not fit is parked lane by lane in vector registers (v_writelane /
  // v_readlane: 431 + 241 of the 2775 vector instructions of the p = 8 Chebyshev form, more at p = 9).  Here every
  // sweep is two phases, one matrix each, and a phase reads its matrix through a pointer the compiler cannot
  // connect with the other phases' (coef_reload: constant address space, so the loads stay scalar): nothing is
  // kept across phases or across bricks, the 41 doubles of a phase fit.
  //   x: W = M u, U = K u;   y: s = K W, W = M W, U = c0 M U + c1 s;   z: W = c2 K W + M U
  // ------------------------------------------------------------------------------------------
  // degrees that run the sliced sweeps (MGX_MACRO_SLICED_FROM: A/B builds)
#ifndef MGX_MACRO_SLICED_FROM
#define MGX_MACRO_SLICED_FROM 7
#endif
  // (not in the fused transfer forms: with the registers of the restriction / interpolation sweeps next to them the
  // sliced sweeps spill -- p = 8 residual + restriction: 412 B per lane)
  template <int P, int MODE>
  constexpr bool kSlicedSweeps = P >= MGX_MACRO_SLICED_FROM && MODE != kResidualRestrict && MODE != kChebFirstProlong;

  template <typename T>
  __device__ __forceinline__ const EOMat<T> &coef_reload(const EOMat<T> *m)
  {
    typedef const EOMat<T> __attribute__((address_space(4))) *ConstPtr;
    ConstPtr q = (ConstPtr)(uintptr_t)m;
    asm volatile("" : "+s"(q));
    return *(const EOMat<T> *)q;
  }

  template <int P, typename T, typename Mark, typename Barrier>
  __device__ __forceinline__ void brick_sweeps_sliced(int tid, T *__restrict__ U, T *__restrict__ W, const Basis1D<T> *__restrict__ B,
                                                      T c0, T c1, T c2, Mark mark, Barrier barrier)
  {
    using C           = MCfg<P, T>;
    constexpr int G   = C::G, NT = C::THREADS, LINES = C::LINES, N = P + 1, NB = C::NB;
    constexpr int REM = LINES - NT;
    const bool    ln  = tid < LINES, rm = REM > 0 && tid < REM * NB;
    const int     rl  = NT + tid / NB, rc = tid % NB; // cell-split pass: line and cell block of this lane
    // seam of the cell-split pass: the block's first output belongs to the previous block's lane
    auto seam = [&](T(&y)[N]) {
      const T yn = next_lane(y[0]);
      if (rc + 1 < NB)
        y[P] += yn;
    };
    // (the lines of the cell-split pass after the whole lines, in phases of their own: their values are not live
    // next to a whole line's)
    // ---- x ----
    if (ln)
      {
        T in[G], o[G];
#pragma unroll
        for (int j = 0; j < G; ++j)
          in[j] = U[tid * G + j];
#pragma unroll
        for (int ph = 0; ph < 2; ++ph)
          {
            const EOMat<T> &A   = coef_reload(ph == 0 ? &B->mass : &B->lapl);
            T              *dst = ph == 0 ? W : U;
            macro_apply<P, T>(A, in, o);
#pragma unroll
            for (int j = 0; j < G; ++j)
              dst[tid * G + j] = o[j];
          }
      }
    if (rm)
      {
        T         seg[N], y[N];
        const int base = rl * G + rc * P;
#pragma unroll
        for (int i = 0; i < N; ++i)
          seg[i] = U[base + i];
#pragma unroll
        for (int ph = 0; ph < 2; ++ph)
          {
            const EOMat<T> &A   = coef_reload(ph == 0 ? &B->mass : &B->lapl);
            T              *dst = ph == 0 ? W : U;
            cell_apply<P, T>(A, seg, y);
            seam(y);
#pragma unroll
            for (int i = 1; i < N; ++i)
              dst[base + i] = y[i];
            if (rc == 0)
              dst[base] = y[0];
          }
      }
    barrier();
    mark(0);
    // ---- y ----
    if (ln)
      {
        T         a[G], s2[G], t2[G];
        const int lb = (tid / G) * (G * G) + tid % G;
#pragma unroll
        for (int j = 0; j < G; ++j)
          a[j] = W[lb + j * G];
        macro_apply<P, T>(coef_reload(&B->lapl), a, s2);
        const EOMat<T> &M = coef_reload(&B->mass);
        macro_apply<P, T>(M, a, t2);
#pragma unroll
        for (int j = 0; j < G; ++j)
          W[lb + j * G] = t2[j];
#pragma unroll
        for (int j = 0; j < G; ++j)
          a[j] = U[lb + j * G];
        macro_apply<P, T>(M, a, t2);
#pragma unroll
        for (int j = 0; j < G; ++j)
          U[lb + j * G] = fma(c0, t2[j], c1 * s2[j]);
      }
    if (rm)
      {
        T         seg[N], y[N], z[N], r[N];
        const int base = (rl / G) * (G * G) + rl % G + rc * P * G;
#pragma unroll
        for (int i = 0; i < N; ++i)
          seg[i] = W[base + i * G];
        cell_apply<P, T>(coef_reload(&B->lapl), seg, z);
        const EOMat<T> &M = coef_reload(&B->mass);
        cell_apply<P, T>(M, seg, y);
#pragma unroll
        for (int i = 0; i < N; ++i)
          seg[i] = U[base + i * G];
        cell_apply<P, T>(M, seg, r);
#pragma unroll
        for (int i = 0; i < N; ++i)
          z[i] = fma(c0, r[i], c1 * z[i]);
        seam(y);
        seam(z);
#pragma unroll
        for (int i = 1; i < N; ++i)
          {
            W[base + i * G] = y[i];
            U[base + i * G] = z[i];
          }
        if (rc == 0)
          {
            W[base] = y[0];
            U[base] = z[0];
          }
      }
    barrier();
    mark(1);
    // ---- z ----
    if (ln)
      {
        T a[G], r[G], o[G];
#pragma unroll
        for (int j = 0; j < G; ++j)
          a[j] = W[tid + j * (G * G)];
        macro_apply<P, T>(coef_reload(&B->lapl), a, r);
#pragma unroll
        for (int j = 0; j < G; ++j)
          a[j] = U[tid + j * (G * G)];
        macro_apply<P, T>(coef_reload(&B->mass), a, o);
#pragma unroll
        for (int j = 0; j < G; ++j)
          W[tid + j * (G * G)] = fma(c2, r[j], o[j]);
      }
    if (rm)
      {
        T         seg[N], y[N], q[N];
        const int base = rl + rc * P * (G * G);
#pragma unroll
        for (int i = 0; i < N; ++i)
          seg[i] = W[base + i * (G * G)];
        cell_apply<P, T>(coef_reload(&B->lapl), seg, q);
#pragma unroll
        for (int i = 0; i < N; ++i)
          seg[i] = U[base + i * (G * G)];
        cell_apply<P, T>(coef_reload(&B->mass), seg, y);
#pragma unroll
        for (int i = 0; i < N; ++i)
          y[i] = fma(c2, q[i], y[i]);
        seam(y);
#pragma unroll
        for (int i = 1; i < N; ++i)
          W[base + i * (G * G)] = y[i];
        if (rc == 0)
          W[base] = y[0];
      }
  }

  // item table word: bits 0..9 entity slot of the brick, 10..22 brick point, 23..31 offset in the entity
  __device__ __forceinline__ uint32_t item_slot(uint32_t m) { return m & 1023u; }
  __device__ __forceinline__ uint32_t item_point(uint32_t m) { return (m >> 10) & 8191u; }
  __device__ __forceinline__ uint32_t item_offset(uint32_t m) { return m >> 23; }

  // the fused Chebyshev forms (they use the inverse diagonal and keep the source value)
  __host__ __device__ constexpr bool is_cheb_mode(int mode)
  {
    return (mode >= kCheb && mode <= kChebOldInit) || mode == kChebFirstProlong;
  }

  // Vector access through buffer descriptors: address = (scalar base) + (32-bit byte offset in one
  // VGPR), no 64-bit address arithmetic and no VGPR pair per access in flight; offsets at or beyond
  // the vector's size are out of range: such a load returns zero and such a store is dropped
  // without touching memory, which is how constrained / masked items are handled (kOob).  The host
  // only selects this kernel for vectors below 4 GB.
  using rsrc_t = __amdgpu_buffer_rsrc_t;
  typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
  // (16-byte aligned, so that no dword of a 16-byte access wraps around to a small offset)
  constexpr uint32_t kOob = 0xFFFFFFF0u;
  // cache-policy bits of a buffer access: 2 = nt (streaming).  Measured on the write-out stores of the
  // finest level: plain form 106.1 -> 103.5 us per colour launch, fused Chebyshev forms 151 -> 156 us
  // (their stores compete with four read streams); on the gather loads: slower everywhere.
  constexpr int kAuxNt = 2;
  __device__ __forceinline__ rsrc_t make_rsrc(const void *p, uint32_t bytes)
  {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
  }
  __device__ __forceinline__ double buf_ld(rsrc_t r, uint32_t off, double)
  {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0));
  }
  __device__ __forceinline__ float buf_ld(rsrc_t r, uint32_t off, float)
  {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
  }
  template <int AUX = 0>
  __device__ __forceinline__ void buf_st(rsrc_t r, uint32_t off, double v)
  {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, v), r, off, 0, AUX);
  }
  template <int AUX = 0>
  __device__ __forceinline__ void buf_st(rsrc_t r, uint32_t off, float v)
  {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v), r, off, 0, AUX);
  }

  // two adjacent values with one 16-byte (fp64) / 8-byte (fp32) access per lane
  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
  __device__ __forceinline__ void buf_ld2(rsrc_t r, uint32_t off, double &a, double &b)
  {
    const u32x4_t v = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
    a               = __builtin_bit_cast(double, u32x2_t{v.x, v.y});
    b               = __builtin_bit_cast(double, u32x2_t{v.z, v.w});
  }
  __device__ __forceinline__ void buf_ld2(rsrc_t r, uint32_t off, float &a, float &b)
  {
    // The two dwords are taken from one 64-bit integer: read as elements 0 and 1 of the vector the
    // builtin returns, hipcc 7.2 narrows the access to ONE dword load and hands the same dword to
    // both (tools/experiments/buffer_probe.hip shows the hardware side is fine).
    const unsigned long long u = __builtin_bit_cast(unsigned long long, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0));
    a                          = __builtin_bit_cast(float, (unsigned int)u);
    b                          = __builtin_bit_cast(float, (unsigned int)(u >> 32));
  }
  __device__ __forceinline__ void buf_st2(rsrc_t r, uint32_t off, double a, double b)
  {
    const u32x2_t lo = __builtin_bit_cast(u32x2_t, a), hi = __builtin_bit_cast(u32x2_t, b);
    __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{lo.x, lo.y, hi.x, hi.y}, r, off, 0, 0);
  }
  __device__ __forceinline__ void buf_st2(rsrc_t r, uint32_t off, float a, float b)
  {
    __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{__builtin_bit_cast(unsigned int, a), __builtin_bit_cast(unsigned int, b)}, r,
                                          off, 0, 0);
  }

  template <typename T>
  struct PostRsrc
  {
    rsrc_t a, b, old, out, partial, src, srcw, xw, priv;
  };

  // The fused post-operation on one assembled value (what the reference passes as
  // operation_after_loop, laplace_operator.h:723-741), split into the part that issues the loads
  // and the part that consumes them, so that the loads of the next chunk of items can be in flight
  // while a chunk is computed and stored.
  // operands of one unit of the write-out: a pair of adjacent DoFs of one entity ([0], [1]) or a
  // single DoF ([0])
  template <typename T>
  struct PostOps
  {
    T pv[2], av[2], bv[2], ov[2]; // partial sum of earlier launches, operands of the post-operation
  };

  // w = entity table word (bit 30 FIRST, bit 31 LAST), off = byte offset of the (first) DoF
  template <typename T, int MODE, bool DTAB, bool PAIR>
  __device__ __forceinline__ void post_issue(const PostRsrc<T> &R, uint32_t w, uint32_t off, PostOps<T> &o)
  {
    const bool valid        = w != kInvalid;
    // (kResidualRestrict hands nothing over between the bricks, see post_finish)
    const bool need_partial = valid && !(w & 0x40000000u) && MODE != kResidualRestrict; // not FIRST
    const bool last         = valid && (w >> 31);
    o.pv[0] = o.pv[1] = o.av[0] = o.av[1] = o.bv[0] = o.bv[1] = o.ov[0] = o.ov[1] = T(0);
    const uint32_t ol = last ? off : kOob, op = need_partial ? off : kOob;
    auto           ld = [&](rsrc_t r, uint32_t f, T(&v)[2]) {
      if (PAIR)
        buf_ld2(r, f, v[0], v[1]);
      else
        v[0] = buf_ld(r, f, T());
    };
    // partial sums of earlier colour launches exist on the brick surface only: whole waves of
    // interior items skip the load
#ifndef MGX_MACRO_NOCARRIER // diagnostic build (wrong results): what the launches cost without the carrier traffic
    if (__builtin_amdgcn_ballot_w64(need_partial) != 0)
      ld(R.partial, op, o.pv);
#endif
    if (MODE != kPlain)
      ld(R.a, ol, o.av);
    if (is_cheb_mode(MODE) && !DTAB)
      ld(R.b, ol, o.bv);
    if (MODE == kCheb || MODE == kCgUpdate)
      ld(R.old, ol, o.ov);
    if (MODE == kCgUpdate)
      ld(R.src, ol, o.bv); // p_old, for x += alpha p_old
  }

  // bv = inverse diagonal at this DoF (loaded with the operands, or from the per-item table)
  template <typename T, int MODE>
  __device__ __forceinline__ T post_finish(const BrickPost<T> &post, T pv, T av, T ov, T bv, bool last, T val, T xi)
  {
    val += pv; // out-of-range loads returned zero
    if (MODE == kPlain || MODE == kCgUpdate)
      return val;
    else if (MODE == kResidualRestrict)
      // The restriction is linear: R (b - A x) = sum over the bricks of R_brick (b_brick - (A x)_brick), with
      // (A x)_brick the brick's own partial sums on ALL its points and b_brick = b on the points the brick
      // completes (every DoF has exactly one LAST visitor).  No partial sum travels between the bricks, nothing is
      // stored on the fine level; on a decomposed mesh the interface DoFs (never LAST) miss only their b, which the
      // owner adds (launch_interface_restrict).
      return (last ? av : T(0)) - val;
    else if (MODE == kResidual)
      return last ? av - val : val;
    else
      {
        if (MODE == kChebOldInit)
          ov = post.f0 * bv * av; // the x_1 of the previous iteration, recomputed (bitwise the same)
        T xn = xi + post.f2 * bv * (av - val);
        if (MODE == kCheb || MODE == kChebOldInit)
          xn += post.f1 * (xi - ov);
        else if (MODE == kChebZeroOld || MODE == kChebInit)
          xn += post.f1 * xi;
        return last ? xn : val;
      }
  }
} // namespace mgx
