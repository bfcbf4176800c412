// mgx_macro2.hip -- second pipeline of the macro-element brick loop for the forms whose write-out loads nothing but
// partial sums: plain (LaplaceOperator::vmult, laplace_operator.h:527-601) and residual (vmult_residual, :605).  Same
// operator, same sweeps, same entity tables and the same bits as mgx_macro.hip.
//
// What round 4 measured on the first pipeline (profiles/r04_*, 135 M DoFs, p = 4):
//   * phase stamps of the plain form: a brick costs a workgroup 24 100 cycles -- sweeps 10 500, tables parked + gather
//     issued 3 100, write-out 5 500 (of which most is WAITING: the loads of the partial sums are issued behind the gather
//     of the next brick, loads return in order, so every store waits for the whole gather), landing the gather 2 000;
//   * a build without the carrier accesses (wrong results) ran 12 % faster, an idealised stream of the same bytes with
//     the same residency (tools/experiments/stream_probe.hip) at 6.2 TB/s against the kernel's 3.7;
//   * counters: waves spend 50 % of their cycles unable to issue (SQ_WAIT_INST_ANY; 23 % on the LDS queue), the LDS
//     array is busy 43-63 % of the launch with 30 % of that bank conflicts of the point scatter, the vector-memory
//     queues are full a third of the time (SQ_VMEM_TA_*_FIFO_FULL, TCP_PENDING_STALL 55 %): the CU-internal data
//     paths bound the kernel, not HBM and not occupancy (a 512-thread form with four waves per SIMD ran no faster).
// Hence this pipeline removes waits and instructions rather than bytes:
//   * the item table lists the (G - 2)^3 interior points of the brick first (build_item_map2): only the last few value
//     slots of a thread (7 of 20 at p = 4) can lie on the brick surface, i.e. carry partial sums, be constrained or sit
//     on a rank interface; the interior slots need neither flags nor carrier accesses;
//   * the partial sums of the surface slots are requested BEFORE the sweeps and have arrived when the write-out starts:
//     it waits for nothing (residual: only for its right-hand side);
//   * the entity table of the current brick has LDS of its own (the 2.9 kB that two arrays of 17^3 doubles leave of a
//     workgroup's 80 kB): no parking of the current table, one barrier less per brick.
// Measured: plain 104 -> 95 us per colour launch.  Also measured and not kept (tools/experiments/
// r04_macro2_wide_early_variants.hip.txt): the gather of the next brick issued before the sweeps (its loads then queue
// behind the stores of the write-out just before: 99 us); 512-thread workgroups with the sweeps as cell-block tasks
// (four waves per SIMD: 98 us, the LDS array 63 % busy); the Chebyshev forms on this pipeline (143-148 against 144 us:
// their write-out waits for its operands either way).
// Barriers order LDS traffic only (lds_barrier): global loads stay in flight across them.
#include "mgx_macro_device.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#ifndef MGX_MACRO_T
#define MGX_MACRO_T double
#endif
// value slots per chunk of the write-out of the Chebyshev forms (operands of chunk c + 1 requested before chunk c is
// computed and stored)
#ifndef MGX_MACRO2_CHUNK
#define MGX_MACRO2_CHUNK 3
#endif

#ifdef MGX_MACRO_STAMPS
// Diagnostic build only (make stamps): thread 0 of every workgroup records s_memtime at the phase boundaries of its
// fourth brick; tools/macro_stamps.py reads them back.  Never compiled into the production library.
#ifndef MGX_MACRO_STAMP_MODE
#define MGX_MACRO_STAMP_MODE -1 // >= 0: only launches of this BrickMode leave stamps
#endif
__device__ unsigned long long g_mgx_stamps2[8192 * 16];
#define MGX_STAMP(k)                                                                                              \
  do                                                                                                              \
    {                                                                                                             \
      if (threadIdx.x == 0 && blockIdx.x < 8192 && (MGX_MACRO_STAMP_MODE < 0 || MODE == MGX_MACRO_STAMP_MODE))      \
        g_mgx_stamps2[blockIdx.x * 16 + (k)] = ((k) == 15 || (k) == 13) ? __builtin_amdgcn_s_memrealtime()         \
                                                                         : __builtin_amdgcn_s_memtime();           \
    }                                                                                                             \
  while (0)
#if MGX_MACRO_IS_F64
extern "C" int mgx_debug_read_stamps2(unsigned long long *host, int n_blocks)
{
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_mgx_stamps2), sizeof(unsigned long long) * 16 * n_blocks);
}
#endif
#define MGX_STAMP_IT(k)  \
  do                     \
    {                    \
      if (mgx_iter == 3) \
        MGX_STAMP(k);    \
    }                    \
  while (0)
#else
#define MGX_STAMP(k) ((void)0)
#define MGX_STAMP_IT(k) ((void)0)
#endif

namespace mgx
{
  template <int P, typename T>
  struct M2Cfg
  {
    using C = MCfg<P, T>;
    static constexpr int G = C::G, NE = C::NE, NB = C::NB, NPTS = C::NPTS, LINES = C::LINES, THREADS = C::THREADS;
    static constexpr int IT    = (NPTS + THREADS - 1) / THREADS;   // value slots per thread
    static constexpr int NINT  = (G - 2) * (G - 2) * (G - 2);      // points in the interior of the brick: listed first
    static constexpr int JINT  = NINT / THREADS;                   // slots below this one are interior on every lane
    static constexpr int JSURF = IT - JINT;                        // slots that may lie on the brick surface
    static constexpr int TW    = (NE * 4 + (int)sizeof(T) - 1) / (int)sizeof(T); // one entity table, in T's
    static constexpr int ASZ   = NPTS > TW ? NPTS : TW;            // (low degrees: U must hold a parked table)
    static constexpr int LDS   = 2 * ASZ * (int)sizeof(T) + NE * 4;
    static constexpr int WG_LDS  = 163840 / LDS;
    static constexpr int WG_WAVE = 32 / (THREADS / 64);
    static constexpr int WGS     = WG_LDS < WG_WAVE ? (WG_LDS < 8 ? WG_LDS : 8) : (WG_WAVE < 8 ? WG_WAVE : 8);
    static constexpr int WAVES   = WGS * (THREADS / 64);
    // registers: three lines of the sweeps, the partial sums in flight, the values kept across the sweeps by the
    // Chebyshev forms, item words.  (Without the IT values in the estimate p = 7 was held to 168 registers for a third
    // workgroup per CU and spilled 28 ... 212 B per lane: residual form 223 us, old-from-rhs form 247 us per colour
    // launch of 64^3 cells against 115 us of the plain form.)
    // p = 7, 8: the line blocks of the sweeps (N = 8, 9 values, their even-odd halves and results) on top -- fp32 p = 8
    // held to 168 registers spilled 52 B per lane in the old-from-rhs form: 118 against 92 us per colour launch
    static constexpr int REGS = (3 * G + JSURF + IT + ((P == 7 || P == 8) ? 4 * (P + 1) : 0)) * ((int)sizeof(T) / 4) + IT + 40;
#ifdef MGX_M2_RMAX // A/B builds
    static constexpr int RMAX = MGX_M2_RMAX;
#else
    static constexpr int RMAX = REGS > 168 ? 2 : (REGS > 128 ? 3 : 4);
#endif
    static constexpr int MINW = (WAVES + 3) / 4 < RMAX ? (WAVES + 3) / 4 : RMAX;
  };

  // The three sweeps of the macro-element operator on the brick arrays (see the head of mgx_macro.hip):
  //   x: U -> (W = Mb u, U = Kb u);  y: (W, U) -> (W = Mb W, U = c0 Mb U + c1 Kb W);  z: W = Mb U + c2 Kb W
  // One thread per line, the REM lines beyond the thread count as one cell block per thread of the first waves.
  // LDS-only barriers after the x and the y sweep (mark(0), mark(1) behind them: phase stamps of the diagnostic
  // build); the caller synchronises after the z sweep.
  template <int P, typename T, typename Mark>
  __device__ __forceinline__ void brick_sweeps(int tid, T *__restrict__ U, T *__restrict__ W, const EOMat<T> &M, const EOMat<T> &K,
                                               T c0, T c1, T c2, Mark mark)
  {
    using C             = MCfg<P, T>;
    constexpr int G     = C::G, NT = C::THREADS, LINES = C::LINES, N = P + 1, NB = C::NB;
    constexpr int REM   = LINES - NT;
    if (tid < LINES)
      {
        const int l = tid;
        T         in[G], t1[G], k1[G];
#pragma unroll
        for (int j = 0; j < G; ++j)
          in[j] = U[l * G + j];
        macro_apply2<P, T>(M, K, in, t1, k1);
#pragma unroll
        for (int j = 0; j < G; ++j)
          {
            W[l * G + j] = t1[j];
            U[l * G + j] = k1[j];
          }
      }
    if (REM > 0 && tid < REM * NB)
      {
        const int l = NT + tid / NB, c = tid % NB, base = l * G + c * P;
        T         seg[N], y[N], z[N];
#pragma unroll
        for (int i = 0; i < N; ++i)
          seg[i] = U[base + i];
        cell_apply2<P, T>(M, K, seg, y, z);
        const T yn = next_lane(y[0]), zn = next_lane(z[0]);
        if (c + 1 < NB)
          {
            y[P] += yn;
            z[P] += zn;
          }
#pragma unroll
        for (int i = 1; i < N; ++i)
          {
            W[base + i] = y[i];
            U[base + i] = z[i];
          }
        if (c == 0)
          {
            W[base] = y[0];
            U[base] = z[0];
          }
      }
    lds_barrier();
    mark(0);
    if (REM > 0 && tid < REM * NB)
      {
        const int l = NT + tid / NB, c = tid % NB, base = (l / G) * (G * G) + l % G + c * P * G;
        T         seg[N], y[N], z[N], r[N];
#pragma unroll
        for (int i = 0; i < N; ++i)
          seg[i] = W[base + i * G];
        cell_apply2<P, T>(M, K, seg, y, z);
#pragma unroll
        for (int i = 0; i < N; ++i)
          seg[i] = U[base + i * G];
        cell_apply<P, T>(M, seg, r);
#pragma unroll
        for (int i = 0; i < N; ++i)
          z[i] = fma(c0, r[i], c1 * z[i]);
        const T yn = next_lane(y[0]), zn = next_lane(z[0]);
        if (c + 1 < NB)
          {
            y[P] += yn;
            z[P] += zn;
          }
#pragma unroll
        for (int i = 1; i < N; ++i)
          {
            W[base + i * G] = y[i];
            U[base + i * G] = z[i];
          }
        if (c == 0)
          {
            W[base] = y[0];
            U[base] = z[0];
          }
      }
    if (tid < LINES)
      {
        const int l    = tid;
        const int base = (l / G) * (G * G) + l % G;
        T         a[G], t2[G], s2[G];
#pragma unroll
        for (int j = 0; j < G; ++j)
          a[j] = W[base + j * G];
        macro_apply2<P, T>(M, K, a, t2, s2);
#pragma unroll
        for (int j = 0; j < G; ++j)
          W[base + j * G] = t2[j];
#pragma unroll
        for (int j = 0; j < G; ++j)
          a[j] = U[base + j * G];
        macro_apply<P, T>(M, a, t2);
#pragma unroll
        for (int j = 0; j < G; ++j)
          U[base + j * G] = fma(c0, t2[j], c1 * s2[j]);
      }
    lds_barrier();
    mark(1);
    if (REM > 0 && tid < REM * NB)
      {
        const int l = NT + tid / NB, c = tid % NB, base = l + c * P * (G * G);
        T         seg[N], y[N], r[N];
#pragma unroll
        for (int i = 0; i < N; ++i)
          seg[i] = W[base + i * (G * G)];
        cell_apply<P, T>(K, seg, r);
#pragma unroll
        for (int i = 0; i < N; ++i)
          seg[i] = U[base + i * (G * G)];
        cell_apply<P, T>(M, seg, y);
#pragma unroll
        for (int i = 0; i < N; ++i)
          y[i] = fma(c2, r[i], y[i]);
        const T yn = next_lane(y[0]);
        if (c + 1 < NB)
          y[P] += yn;
#pragma unroll
        for (int i = 1; i < N; ++i)
          W[base + i * (G * G)] = y[i];
        if (c == 0)
          W[base] = y[0];
      }
    if (tid < LINES)
      {
        const int l = tid;
        T         a[G], r[G], o[G];
#pragma unroll
        for (int j = 0; j < G; ++j)
          a[j] = W[l + j * (G * G)];
        macro_apply<P, T>(K, a, r);
#pragma unroll
        for (int j = 0; j < G; ++j)
          a[j] = U[l + j * (G * G)];
        macro_apply<P, T>(M, a, o);
#pragma unroll
        for (int j = 0; j < G; ++j)
          W[l + j * (G * G)] = fma(c2, r[j], o[j]);
      }
  }

  // (residual + restriction at p >= 5, where a brick is the 8 children of one parent: the 40 registers of the right-hand
  // side in flight across the sweeps cost more than the wait they remove -- 274 against 257 us per colour launch at p = 8)
  // Of the Chebyshev forms the two that never store the first iterate (kChebInit, kChebOldInit) run here, with the inverse
  // diagonal taken from the per-item table (uniform meshes; where it has to be streamed they stay on the first
  // pipeline): 119 -> 116 and 133 -> 126 us per colour launch.  The general iteration (kCheb: 144 -> 148 us) and its
  // two special cases gain nothing from the earlier partial sums -- their write-out waits for b and x_old either way --
  // and stay on the first pipeline.
  __host__ __device__ constexpr bool macro2_covers(int mode, int p = 4)
  {
    return mode == kPlain || mode == kResidual || ((mode == kResidualRestrict || mode == kChebInit) && p <= 4) || mode == kChebOldInit;
  }

  template <int P, typename T, int MODE>
  __global__ void __launch_bounds__((M2Cfg<P, T>::THREADS), (M2Cfg<P, T>::MINW))
    brick_macro2_kernel(const T *__restrict__ src, uint32_t brick_first, uint32_t brick_count,
                        const uint32_t *__restrict__ ent_base, const uint32_t *__restrict__ item_map,
                        const Basis1D<T> *__restrict__ B, T c0, T c1, T c2, BrickPost<T> post, uint32_t vec_bytes)
  {
    static_assert(macro2_covers(MODE, (MODE == kResidualRestrict || MODE == kChebInit) ? 4 : P), "form not covered by this pipeline");
    using C            = M2Cfg<P, T>;
    constexpr int NT   = C::THREADS, IT = C::IT, JINT = C::JINT, JSURF = C::JSURF, NE = C::NE;
    constexpr int NEW  = (NE + NT - 1) / NT; // entity words per thread
    __shared__ T        U[C::ASZ];
    __shared__ T        W[C::ASZ];
    __shared__ uint32_t E[NE]; // entity table of the current brick

    const int tid = threadIdx.x;
    uint32_t  b   = blockIdx.x;
    if (b >= brick_count)
      return;
    MGX_STAMP(0);
    MGX_STAMP(15);
#ifdef MGX_MACRO_STAMPS
    int mgx_iter = 0;
#endif
    auto live = [&](int j) { return (j + 1) * NT <= C::NPTS || tid + j * NT < C::NPTS; };
    uint32_t mw[IT]; // item words: the same for every brick
#pragma unroll
    for (int j = 0; j < IT; ++j)
      mw[j] = item_map[live(j) ? tid + j * NT : 0];
    const rsrc_t rsrc = make_rsrc(src, vec_bytes), r_a = make_rsrc(post.a, vec_bytes), r_out = make_rsrc(post.out, vec_bytes),
                 r_partial = make_rsrc(post.partial, vec_bytes), r_old = make_rsrc(post.old, vec_bytes);
    const EOMat<T> &M = B->mass, &K = B->lapl;
    // the fused Chebyshev forms keep the source value of every item for the update and the inverse diagonal of the
    // thread's items (post.b = the per-item table in the order of this pipeline's item map) for all bricks
    constexpr bool kKeepX = is_cheb_mode(MODE);
    T              dv[kKeepX ? IT : 1];
    if (kKeepX)
      {
#pragma unroll
        for (int j = 0; j < IT; ++j)
          dv[kKeepX ? j : 0] = post.b[live(j) ? tid + j * NT : 0];
      }

    uint32_t en[NEW] = {}; // entity table words of the next brick
    auto     table_load = [&](uint32_t brick, uint32_t(&e)[NEW]) {
#pragma unroll
      for (int j = 0; j < NEW; ++j)
        {
          const int i = tid + j * NT;
          e[j]        = ent_base[(size_t)(brick_first + brick) * NE + (i < NE ? i : 0)];
        }
    };
    auto table_store = [&](uint32_t *dst, const uint32_t(&e)[NEW]) {
#pragma unroll
      for (int j = 0; j < NEW; ++j)
        if (tid + j * NT < NE)
          dst[tid + j * NT] = e[j];
    };
    // byte offset of a DoF; constrained entity: out of range (loads return zero, stores are dropped --
    // vector_access_reduced.h:174-179, 431-433)
    auto unit_offset = [&](uint32_t w, uint32_t m) {
      return w != kInvalid ? (ent_index(w) + item_offset(m)) * (uint32_t)sizeof(T) : kOob;
    };
    T    g[IT];
    auto gather_issue = [&](const uint32_t *tab) {
#pragma unroll
      for (int j = 0; j < IT; ++j)
        {
          g[j] = T(0);
          if (live(j)) // (kChebInit gathers the right-hand side: x_1 = f0 D^-1 b is formed while landing)
            g[j] = buf_ld(MODE == kChebInit ? r_a : rsrc, unit_offset(tab[item_slot(mw[j])], mw[j]), T());
        }
    };
    T    xs[kKeepX ? IT : 1];
    auto gather_land = [&]() {
#pragma unroll
      for (int j = 0; j < IT; ++j)
        {
          T v = g[j];
          if (MODE == kChebInit)
            v = post.f0 * dv[kKeepX ? j : 0] * g[j];
          if (live(j))
            U[item_point(mw[j])] = v;
          if (kKeepX)
            xs[kKeepX ? j : 0] = v;
        }
    };
    // partial sums of the surface slots of the current brick (not FIRST: an earlier colour launch left a sum).  The
    // residual + restriction form hands nothing over between the bricks (linear form, see post_finish in
    // mgx_macro_device.hpp)
    constexpr bool kCarrier = MODE != kResidualRestrict;
    T              pp[JSURF];
    auto           partial_issue = [&]() {
#pragma unroll
      for (int j = JINT; j < IT; ++j)
        {
          pp[j - JINT] = T(0);
          if (!kCarrier)
            continue;
#ifndef MGX_MACRO_NOCARRIER // diagnostic build (wrong results): what the launches cost without the carrier traffic
          const uint32_t w    = live(j) ? E[item_slot(mw[j])] : kInvalid;
          const bool     need = w != kInvalid && !(w & 0x40000000u);
          if (__builtin_amdgcn_ballot_w64(need) != 0) // whole waves of FIRST items skip the load
            pp[j - JINT] = buf_ld(r_partial, need ? unit_offset(w, mw[j]) : kOob, T());
#endif
        }
    };

    // right-hand side at the DoFs the current brick completes (residual forms), requested before the sweeps as well
    constexpr bool kRhs = MODE == kResidual || MODE == kResidualRestrict;
    T              av[kRhs ? IT : 1];
    auto           rhs_issue = [&]() {
      if (kRhs)
        {
#pragma unroll
          for (int j = 0; j < IT; ++j)
            {
              const uint32_t w    = live(j) ? E[item_slot(mw[j])] : kInvalid;
              const bool     last = w != kInvalid && (j < JINT || (w >> 31));
              av[kRhs ? j : 0]    = buf_ld(r_a, last ? unit_offset(w, mw[j]) : kOob, T());
            }
        }
    };

    // ---- prologue: table and source of the first brick, table of the second ----
    {
      uint32_t e0[NEW];
      table_load(b, e0);
      if (b + gridDim.x < brick_count)
        table_load(b + gridDim.x, en);
      table_store(E, e0);
      lds_barrier();
      MGX_STAMP(1);
      gather_issue(E);
      gather_land();
      MGX_STAMP(2);
      lds_barrier();
    }

    for (;;)
      {
        const uint32_t bn = b + gridDim.x;
        const bool     has_next = bn < brick_count;
        MGX_STAMP_IT(3);
#ifdef MGX_MACRO_STAMPS
        if (mgx_iter == 4)
          MGX_STAMP(11);
#endif
        partial_issue(); // in flight during the sweeps
        rhs_issue();
        // keep what is derived from the item words (LDS addresses, offsets) out of the registers that live across
        // the sweeps: the compiler must not hoist it out of the brick loop
#pragma unroll
        for (int j = 0; j < IT; ++j)
          asm volatile("" : "+v"(mw[j]));
        if (MODE == kChebInit || MODE == kChebOldInit) // likewise f0 * dv
          {
#pragma unroll
            for (int j = 0; j < IT; ++j)
              asm volatile("" : "+v"(dv[kKeepX ? j : 0]));
          }
        MGX_STAMP_IT(4);
#ifndef MGX_MACRO_NOSWEEP // diagnostic build without the sweeps (wrong results): memory phases alone
        // (this pipeline: from p = 9 on -- its plain form loses 8 % at p = 7 and 2 % at p = 8 with the slices, the
        // old-from-rhs form gains 7 % at p = 9 and nothing below)
        if (kSlicedSweeps<P, MODE> && P >= 9)
          brick_sweeps_sliced<P, T>(
            tid, U, W, B, c0, c1, c2, [&](int k) { MGX_STAMP_IT(5 + k); }, [&]() { lds_barrier(); });
        else
          brick_sweeps<P, T>(tid, U, W, M, K, c0, c1, c2, [&](int k) { MGX_STAMP_IT(5 + k); });
#endif
        lds_barrier();
        MGX_STAMP_IT(7);
        if (has_next)
          {
            // U is free: park the next table there and issue the gather; the write-out runs with it in flight
            table_store(reinterpret_cast<uint32_t *>(U), en);
            lds_barrier();
            gather_issue(reinterpret_cast<const uint32_t *>(U));
          }
        MGX_STAMP_IT(8);

        if constexpr (kKeepX)
          {
            // ---- write-out of the Chebyshev forms: x_new = x + f2 D^-1 (b - A x) [+ f1 (x - x_old)] where the brick
            //      completes the DoF, the partial sum to the carrier elsewhere.  The operands b (and x_old) are requested
            //      in chunks, one chunk ahead of the stores; the partial sums are in registers already ----
            constexpr int kChunk = MGX_MACRO2_CHUNK < IT ? MGX_MACRO2_CHUNK : IT, NCH = (IT + kChunk - 1) / kChunk;
            struct Ops
            {
              T        av, ov;
              uint32_t w, off;
            };
            Ops  ops[2][kChunk];
            auto issue = [&](int c, Ops(&o)[kChunk]) {
#pragma unroll
              for (int k = 0; k < kChunk; ++k)
                {
                  const int j = c * kChunk + k;
                  if (j < IT)
                    o[k].w = live(j) ? E[item_slot(mw[j])] : kInvalid;
                }
#pragma unroll
              for (int k = 0; k < kChunk; ++k)
                {
                  const int j = c * kChunk + k;
                  if (j >= IT)
                    continue;
                  if (j < JINT && o[k].w != kInvalid)
                    o[k].w |= 0xC0000000u;
                  const bool last  = o[k].w != kInvalid && (o[k].w >> 31);
                  o[k].off         = unit_offset(o[k].w, mw[j]);
                  const uint32_t ol = last ? o[k].off : kOob;
                  o[k].av          = buf_ld(r_a, ol, T());
                  o[k].ov          = MODE == kCheb ? buf_ld(r_old, ol, T()) : T(0);
                }
            };
            issue(0, ops[0]);
#pragma unroll
            for (int c = 0; c < NCH; ++c)
              {
                if (c + 1 < NCH)
                  issue(c + 1, ops[(c + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                T val[kChunk];
#pragma unroll
                for (int k = 0; k < kChunk; ++k)
                  {
                    const int j = c * kChunk + k;
                    val[k]      = (j < IT && live(j)) ? W[item_point(mw[j])] : T(0);
                  }
#pragma unroll
                for (int k = 0; k < kChunk; ++k)
                  {
                    const int j = c * kChunk + k;
                    if (j >= IT)
                      continue;
                    const Ops &o   = ops[c & 1][k];
                    const bool vld = o.w != kInvalid, last = vld && (o.w >> 31);
                    const T    res = post_finish<T, MODE>(post, j >= JINT ? pp[j >= JINT ? j - JINT : 0] : T(0), o.av, o.ov,
                                                          dv[kKeepX ? j : 0], last, val[k], xs[kKeepX ? j : 0]);
                    if (j < JINT)
                      buf_st(r_out, o.off, res);
                    else
                      {
                        buf_st(r_out, last ? o.off : kOob, res);
#ifndef MGX_MACRO_NOCARRIER
                        if (__builtin_amdgcn_ballot_w64(vld && !last) != 0)
                          buf_st(r_partial, last ? kOob : o.off, res);
#endif
                      }
                  }
                __builtin_amdgcn_sched_barrier(0);
              }
          }
        else
          {
          // ---- write-out: assembled value (+ partial sum) -> result where the brick completes the DoF (LAST), else
          //      -> carrier.  Every operand is in registers: nothing here waits for memory.  Residual + restriction: the
          //      brick's share of the residual stays in W (rows of constrained DoFs: zero) and is restricted below ----
          {
            uint32_t w[IT];
  #pragma unroll
            for (int j = 0; j < IT; ++j)
              w[j] = live(j) ? E[item_slot(mw[j])] : kInvalid;
  #pragma unroll
            for (int j = 0; j < IT; ++j)
              {
                // interior of the brick: complete after this brick, whatever the schedule says
                if (j < JINT && w[j] != kInvalid)
                  w[j] |= 0xC0000000u;
                const bool     vld = w[j] != kInvalid, last = vld && (w[j] >> 31);
                const uint32_t off = unit_offset(w[j], mw[j]);
                T              val = live(j) ? W[item_point(mw[j])] : T(0);
                if (j >= JINT && kCarrier)
                  val += pp[j >= JINT ? j - JINT : 0]; // (out-of-range loads returned zero)
                if (MODE == kResidualRestrict)
                  {
                    // linear form: b on the points the brick completes minus its own share of A x on all its points
                    if (live(j))
                      W[item_point(mw[j])] = vld ? (last ? av[kRhs ? j : 0] : T(0)) - val : T(0);
                    continue;
                  }
                if (MODE == kResidual && last)
                  val = av[kRhs ? j : 0] - val;
                if (j < JINT)
                  buf_st<kAuxNt>(r_out, off, val);
                else
                  {
                    buf_st<kAuxNt>(r_out, last ? off : kOob, val);
  #ifndef MGX_MACRO_NOCARRIER
                    if (__builtin_amdgcn_ballot_w64(vld && !last) != 0) // whole waves of completed items skip the store
                      buf_st(r_partial, last ? kOob : off, val);
  #endif
                  }
              }
          }
          }
        if (MODE == kResidualRestrict)
          {
            constexpr int CE1 = C::NB + 1, CNP = (C::NB / 2) * P + 1; // coarse entities / points per direction of the parents
            lds_barrier();
            const uint32_t *ctab = post.coarse_blocks + (size_t)(brick_first + b) * (CE1 * CE1 * CE1);
            if (post.coarse_scratch) // uniform
              restrict_brick<P, T, NT, true>(tid, W, B->P1eo, nullptr, ctab,
                                             post.coarse_scratch + (size_t)(brick_first + b) * (CNP * CNP * CNP));
            else
              restrict_brick<P, T, NT, false>(tid, W, B->P1eo, post.coarse, ctab);
          }
        MGX_STAMP_IT(9);
        if (!has_next)
          break;
        lds_barrier(); // everyone is done with W, with the parked table and with the table of this brick
        gather_land();
        table_store(E, en);
        if (bn + gridDim.x < brick_count)
          table_load(bn + gridDim.x, en); // in flight during the sweeps of the next brick
        MGX_STAMP_IT(10);
        lds_barrier();
        b = bn;
#ifdef MGX_MACRO_STAMPS
        ++mgx_iter;
#endif
      }
#ifdef MGX_MACRO_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MGX_STAMP(14);
    MGX_STAMP(13);
#endif
  }

  // ------------------------------------------------------------------------------------------
  static uint32_t macro2_cus(const OperatorData &op)
  {
    static const int cus = [] {
      int dev = 0, n = 256;
      if (hipGetDevice(&dev) == hipSuccess)
        (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
      return std::max(1, n);
    }();
    return op.macro_wg_x16 ? std::max<uint32_t>(1u, (uint32_t)cus * op.macro_wg_x16 / 16u) : (uint32_t)cus;
  }

  template <int P, typename T, int MODE>
  static void macro2_launch(hipStream_t s, const OperatorData &op, const T *src, const BrickPost<T> &post, int g0, int g1)
  {
    using C             = M2Cfg<P, T>;
    const BrickData &bd = op.bricks;
    // kResidualRestrict with a coarse scratch array: the bricks hand nothing to each other and write disjoint
    // addresses -- one launch for all of them
    const bool one_launch = MODE == kResidualRestrict && post.coarse_scratch != nullptr;
    for (int c = g0; c < g1; ++c)
      {
        uint32_t first = bd.colour_start[c], count = bd.colour_start[c + 1] - first;
        if (one_launch)
          {
            if (c != g0)
              break;
            count = bd.colour_start[g1] - first;
          }
        if (count == 0)
          continue;
        // persistent workgroups: as many as are resident at once (WGS per CU)
        const uint32_t grid = std::min<uint32_t>(count, (uint32_t)(op.macro_wg_x16 ? 1 : C::WGS) * macro2_cus(op));
        hipLaunchKernelGGL((brick_macro2_kernel<P, T, MODE>), dim3(grid), dim3(C::THREADS), 0, s, src, first, count, bd.ent_base,
                           bd.item_map2, (const Basis1D<T> *)op.basis, (T)op.coef[0], (T)op.coef[1], (T)op.coef[2], post,
                           (uint32_t)(op.n_dofs * sizeof(T)));
      }
  }

  template <int P, typename T>
  static void macro2_modes(hipStream_t s, const OperatorData &op, int mode, const T *src, const BrickPost<T> &post, int g0, int g1)
  {
    switch (mode)
      {
        case kPlain: macro2_launch<P, T, kPlain>(s, op, src, post, g0, g1); break;
        case kResidual: macro2_launch<P, T, kResidual>(s, op, src, post, g0, g1); break;
        case kChebInit:
          if constexpr (P <= 4) // (p = 8: 155 against 141 us on the first pipeline)
            macro2_launch<P, T, kChebInit>(s, op, src, post, g0, g1);
          break;
        case kChebOldInit: macro2_launch<P, T, kChebOldInit>(s, op, src, post, g0, g1); break;
        case kResidualRestrict:
          if constexpr (P <= 4)
            macro2_launch<P, T, kResidualRestrict>(s, op, src, post, g0, g1);
          break;
        default: break;
      }
  }

#define MGX_CAT2(a, b) a##b
#define MGX_CAT(a, b) MGX_CAT2(a, b)
  // false: form / degree / vector size not covered by this pipeline (the caller uses the first one)
  // kResidualRestrict: `partial` names the per-brick scratch array of the restricted values (or nullptr: added into
  // `coarse` colour by colour), coarse_blocks the coarse entity table of the bricks' parents
  bool MGX_CAT(launch_macro2_loop_, MGX_MACRO_SUFFIX)(hipStream_t s, const OperatorData &op, int mode, const void *src, const void *a,
                                                      void *out, void *partial, void *coarse, const uint32_t *coarse_blocks, int g0,
                                                      int g1, double f1, double f2, double f0, const void *old)
  {
    using T = MGX_MACRO_T;
    if (!macro2_covers(mode, op.p) || !op.bricks.item_map2 || (uint64_t)op.n_dofs * sizeof(T) >= 0xFFFFFFF0ull)
      return false;
    if (is_cheb_mode(mode) && !op.diag_items2) // inverse diagonal not uniform per item: streamed by the first pipeline
      return false;
    BrickPost<T> post{};
    post.b   = (const T *)op.diag_items2;
    post.old = (const T *)old;
    post.f1  = (T)f1;
    post.f2  = (T)f2;
    post.f0  = (T)f0;
    post.a              = (const T *)a;
    post.out            = (T *)out;
    post.partial        = (T *)partial;
    post.coarse         = (T *)coarse;
    post.coarse_blocks  = coarse_blocks;
    post.coarse_scratch = mode == kResidualRestrict ? (T *)partial : nullptr;
    switch (op.p)
      {
#ifdef MGX_MACRO_ONLY_P
        case MGX_MACRO_ONLY_P: macro2_modes<MGX_MACRO_ONLY_P, T>(s, op, mode, (const T *)src, post, g0, g1); break;
#else
        case 1: macro2_modes<1, T>(s, op, mode, (const T *)src, post, g0, g1); break;
        case 2: macro2_modes<2, T>(s, op, mode, (const T *)src, post, g0, g1); break;
        case 3: macro2_modes<3, T>(s, op, mode, (const T *)src, post, g0, g1); break;
        case 4: macro2_modes<4, T>(s, op, mode, (const T *)src, post, g0, g1); break;
        case 5: macro2_modes<5, T>(s, op, mode, (const T *)src, post, g0, g1); break;
        case 6: macro2_modes<6, T>(s, op, mode, (const T *)src, post, g0, g1); break;
        case 7: macro2_modes<7, T>(s, op, mode, (const T *)src, post, g0, g1); break;
        case 8: macro2_modes<8, T>(s, op, mode, (const T *)src, post, g0, g1); break;
        case 9: macro2_modes<9, T>(s, op, mode, (const T *)src, post, g0, g1); break;
#endif
        default: return false;
      }
    return true;
  }
} // namespace mgx
