// mgx_macro.hip -- the production cell loop of the separable (Cartesian, constant coefficient)
// operator in macro-element form: sum factorisation over the whole brick instead of cell by cell.
//
// Why this is the same operator.  On a Cartesian mesh with one constant diagonal coefficient the
// cell matrix is  A_cell = sum_d c_d (M x M x K_d)  with the 1D matrices M = S^T W S and
// K = S^T D^T W D S (mgx_brick.hip, laplace_operator.h:374-387, 447-491).  A brick is a tensor
// product of NB cells per direction, so the sum of its NB^3 cell matrices factorises as well:
//     sum_cells A_cell = c_x (Mb x Mb x Kb) + c_y (Mb x Kb x Mb) + c_z (Kb x Mb x Mb)
// where Mb, Kb are the 1D matrices *assembled* over the NB cells of a line (G = NB p + 1 points,
// block-banded: one dense (p+1)^2 block per cell, neighbouring blocks overlap in one point).
// The brick's G^3 results are therefore three sweeps over G^2 lines of G points,
//     t1 = Mb_x u, k1 = Kb_x u ; t2 = Mb_y t1, s2 = c_x Mb_y k1 + c_y Kb_y t1 ;
//     out = Mb_z s2 + c_z Kb_z t2,
// exactly the sweeps of brick_sep_kernel but on lines that run through the whole brick: a line
// that is shared by the cells on both sides of a face is swept once instead of once per cell
// ((G/(NB (p+1)))^2 = 0.72 of the flops at p = 4), the result comes out assembled (no LDS
// accumulator, no parity rounds, no read-modify-write), and the two LDS transposes are per brick
// instead of per cell (0.6 of the LDS traffic).  Constrained DoFs enter as zeros and their rows are
// never written, as in read_dof_values_compressed / distribute_local_to_global_compressed
// (vector_access_reduced.h:174-179, 431-433).
//
// Schedule of one 256-thread workgroup (p = 4 and p = 8: G = 17, 289 lines):
//   1. gather: the G^3 source values are loaded entity by entity (consecutive work items are
//      consecutive DoFs of one mesh entity = consecutive addresses) through a per-degree item table
//      (item -> entity slot, offset inside the entity, brick point) and written to the LDS array U;
//      every thread keeps the values it loaded in registers: the write-out below uses the same
//      item -> thread mapping, so the fused Chebyshev update has its x_i without a second read.
//   2. x sweep, thread = line (y,z): reads its line of U, writes Mb u to W and Kb u back to its own
//      line of U (in place: no other thread touches that line in this phase).  The 33 lines beyond
//      the 256 threads are swept in a second pass, one cell block of a line per thread.
//   3. y sweep, thread = line (x,z): reads its lines of W and U, writes t2 / s2 back in place.
//   4. z sweep, thread = line (x,y): reads its lines of W and U, writes the result to W.
//   5. write-out entity by entity with the fused post-operation (BrickMode), in chunks whose loads
//      are issued one chunk ahead of the stores.
// One workgroup barrier between the phases.  LDS: two fp64 arrays of 17^3 and nothing else
// (78.6 kB): two workgroups per CU.  The workgroups are persistent and software-pipelined over
// the bricks of a colour launch -- see brick_macro_kernel below.
#include "mgx_macro_device.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

// units (pairs of DoFs) per chunk of the write-out
#ifndef MGX_MACRO_CHUNK
#define MGX_MACRO_CHUNK (MGX_MACRO_PAIRS ? 3 : 4)
#endif
#ifndef MGX_MACRO_CHUNK_CHEB
#define MGX_MACRO_CHUNK_CHEB (MGX_MACRO_PAIRS ? 2 : 3)
#endif
#ifndef MGX_MACRO_T
#define MGX_MACRO_T double
#endif

#ifdef MGX_MACRO_STAMPS
// Diagnostic build only (make MACROFLAGS=-DMGX_MACRO_STAMPS): thread 0 of every workgroup records
// s_memtime at the phase boundaries; tools/macro_stamps.py reads them back.  Never compiled into
// the production library.
#ifndef MGX_MACRO_STAMP_MODE
#define MGX_MACRO_STAMP_MODE -1 // >= 0: only launches of this BrickMode leave stamps
#endif
__device__ unsigned long long g_mgx_stamps[8192 * 16];
#define MGX_STAMP(k)                                                                    \
  do                                                                                    \
    {                                                                                   \
      if (threadIdx.x == 0 && blockIdx.x < 8192 && (MGX_MACRO_STAMP_MODE < 0 || MODE == MGX_MACRO_STAMP_MODE)) \
        g_mgx_stamps[blockIdx.x * 16 + (k)] = ((k) == 15 || (k) == 13) ? __builtin_amdgcn_s_memrealtime() \
                                                        : __builtin_amdgcn_s_memtime();  \
    }                                                                                   \
  while (0)
extern "C" int mgx_debug_read_stamps(unsigned long long *host, int n_blocks)
{
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_mgx_stamps), sizeof(unsigned long long) * 16 * n_blocks);
}
// stamps inside the brick loop: recorded for the fourth brick of the workgroup only (steady state)
#define MGX_STAMP_IT(k)      \
  do                         \
    {                        \
      if (mgx_iter == 3)     \
        MGX_STAMP(k);        \
    }                        \
  while (0)
#define MGX_STAMP_NEXT() ++mgx_iter
#define mgx_iter_is(k) (mgx_iter == (k))
#else
#define MGX_STAMP(k) ((void)0)
#define MGX_STAMP_IT(k) ((void)0)
#define MGX_STAMP_NEXT() ((void)0)
#define mgx_iter_is(k) false
#endif

namespace mgx
{
  // Persistent, software-pipelined: a workgroup walks over the bricks b = blockIdx.x + j gridDim.x
  // of the colour launch.  While brick b is swept, the entity table of the next brick is in flight
  // (three words per thread); as soon as the z sweep has freed the array U, that table is parked
  // in U's space, the gather of the next brick is issued from it, and the write-out of brick b
  // runs with those loads in flight.  Afterwards the table moves to its own buffer and the
  // gathered values to U.  A brick's two round trips to memory before the first sweep (table,
  // source) are hidden this way; the item table words stay in registers for all bricks.
  //
  // DTAB: the inverse diagonal is not streamed.  On a uniform Cartesian mesh it takes one value per
  // position of a DoF inside the cell period, i.e. per item of the brick (verified against the
  // stored vector when the diagonal is computed, macro_diag_table below); every thread holds the
  // values of its items in registers for all bricks.  One vector stream less in the fused
  // Chebyshev iterations (4 instead of 5 accesses per DoF).
  //
  // FREE (reduced-colour schedules of the plain / residual / Chebyshev forms, BrickData::fr): the eight
  // colour launches exist because a brick hands the partial sums of its surface points to its
  // neighbours through the carrier vector (FIRST / LAST read-modify-writes), and two bricks that
  // share a point must not run at the same time.  Here the entities named by post.surf_off are
  // PRIVATE instead: the brick stores their partial sums in a block of its own (post.priv =
  // [brick][n_surf], entity by entity in item order) and surf_finish (mgx_vector.hip) adds the blocks
  // up per DoF in a fixed order and applies the post-operation there.  Two schedules use it:
  //   one class : every surface entity is private -- ONE launch for the whole level (levels whose
  //               colour launches hold a brick per workgroup or less: nothing to pipeline, every
  //               launch as long as a brick's whole latency chain);
  //   two classes (parity of the brick position): only the entities on brick edges and corners,
  //               which bricks of the same class share, are private (1.1 % of the DoFs at p = 4); the
  //               faces keep the FIRST / LAST hand-over through the carrier between the two launches.
  // The entity table of these schedules (its own brick order and flags) comes with the launch.
  template <int P, typename T, int MODE, bool DTAB, bool FREE = false>
  __global__ void __launch_bounds__((MCfg<P, T>::THREADS), (MCfg<P, T>::MINW))
    brick_macro_kernel(const T *__restrict__ src, uint32_t brick_first, uint32_t brick_count,
                       const uint32_t *__restrict__ ent_base, const uint32_t *__restrict__ item_map,
                       const Basis1D<T> *__restrict__ B, T c0, T c1, T c2, BrickPost<T> post, uint32_t vec_bytes)
  {
    using C              = MCfg<P, T>;
    constexpr int G      = C::G;
    constexpr int NT     = C::THREADS;
    constexpr int IT     = C::IT;
    constexpr int LINES  = C::LINES;
    constexpr int NEW    = (C::NE + NT - 1) / NT; // entity words per thread
    constexpr int N = P + 1, NB = C::NB;
    constexpr int REM    = LINES - NT;            // lines beyond the thread count (cell-split pass)
    // the fused Chebyshev forms need the source value again at write-out time
    constexpr bool kKeepX = is_cheb_mode(MODE) || MODE == kCgUpdate;
    constexpr int  NG     = ((MODE == kChebInit && !DTAB) || MODE == kCgUpdate) ? 2 : 1; // operands gathered per value
    // two arrays of G^3 values and nothing else: 78.6 kB in fp64 at G = 17, two workgroups per CU.
    // The entity table of a brick has no LDS of its own: it is needed before the first sweep (gather)
    // and after the last one (write-out), when one of the arrays is free, and rests in registers
    // (three words per thread) in between.
    // (low degrees: the arrays are padded to hold two parked tables)
    __shared__ T U[C::ASZ];
    __shared__ T W[C::ASZ];

    const int tid = threadIdx.x;
    uint32_t  b   = blockIdx.x;
    if (b >= brick_count)
      return;
    MGX_STAMP(0);
    MGX_STAMP(15);
#ifdef MGX_MACRO_STAGGER // experiment: the second workgroup of a CU (LDS allocation not at 0) starts late
    if (__builtin_amdgcn_s_getreg(6 | (0 << 6) | ((8 - 1) << 11)) != 0)
      for (int i = 0; i < MGX_MACRO_STAGGER; ++i)
        __builtin_amdgcn_s_sleep(64);
#endif
    // Value slots of a thread: pair j of JP (slots 2j, 2j+1 = the two DoFs of pair tid + j NT of the
    // item table) and single j of JS (slot 2 JP + j).  live: the slot exists for this thread.
    constexpr int NP = C::NPAIR, NS = C::NSINGLE, JP = C::JP, JS = C::JS, NU = JP + JS; // NU units
    auto live_p = [&](int j) { return (j + 1) * NT <= NP || tid + j * NT < NP; };
    auto live_s = [&](int j) { return (j + 1) * NT <= NS || tid + j * NT < NS; };
    auto live   = [&](int v) { return v < 2 * JP ? live_p(v / 2) : live_s(v - 2 * JP); };
    // item table words: the same for every brick, resident in registers
    uint32_t mw[IT];
#pragma unroll
    for (int j = 0; j < JP; ++j)
      {
        const int     q = tid + j * NT;
        const u32x2_t m = reinterpret_cast<const u32x2_t *>(item_map)[q < NP ? q : 0];
        mw[2 * j]       = m.x;
        mw[2 * j + 1]   = m.y;
      }
#pragma unroll
    for (int j = 0; j < JS; ++j)
      {
        const int q    = tid + j * NT;
        mw[2 * JP + j] = item_map[2 * NP + (q < NS ? q : 0)];
      }
    T dv[DTAB ? IT : 1]; // inverse diagonal of this thread's items (DTAB: post.b is the table, in item order)
    if (DTAB)
      {
#pragma unroll
        for (int j = 0; j < JP; ++j)
          {
            const int q                = tid + j * NT;
            dv[DTAB ? 2 * j : 0]       = post.b[2 * (q < NP ? q : 0)];
            dv[DTAB ? 2 * j + 1 : 0]   = post.b[2 * (q < NP ? q : 0) + 1];
          }
#pragma unroll
        for (int j = 0; j < JS; ++j)
          {
            const int q               = tid + j * NT;
            dv[DTAB ? 2 * JP + j : 0] = post.b[2 * NP + (q < NS ? q : 0)];
          }
      }
    const rsrc_t      rsrc = make_rsrc(src, vec_bytes);
    const PostRsrc<T> R{make_rsrc(post.a, vec_bytes),   make_rsrc(post.b, vec_bytes),       make_rsrc(post.old, vec_bytes),
                        make_rsrc(post.out, vec_bytes), make_rsrc(post.partial, vec_bytes), rsrc,
                        make_rsrc((MODE == kCgUpdate || MODE == kChebFirstProlong) ? post.src_w : post.out, vec_bytes),
                        make_rsrc(MODE == kCgUpdate ? post.x_w : post.out, vec_bytes),
                        make_rsrc(FREE ? (const void *)post.priv : (const void *)post.out, FREE ? post.priv_bytes : vec_bytes)};
    double cg[4] = {0., 0., 0., 0.}; // kCgUpdate: q.p, r.r, q.r, q.q over the DoFs this workgroup completes
    const EOMat<T>   &M = B->mass, &K = B->lapl;

    uint32_t ec[NEW], en[NEW]; // entity table words of the current / the next brick
    uint32_t sc[FREE ? NEW : 1]; // FREE: the surface table, parked next to the entity tables for the write-out
    if (FREE)
      {
#pragma unroll
        for (int j = 0; j < NEW; ++j)
          {
            const int i      = tid + j * NT;
            sc[FREE ? j : 0] = post.surf_off[i < C::NE ? i : 0];
          }
      }
    auto table_load = [&](uint32_t brick, uint32_t(&e)[NEW]) {
#pragma unroll
      for (int j = 0; j < NEW; ++j)
        {
          const int i = tid + j * NT;
          e[j]        = ent_base[(size_t)(brick_first + brick) * C::NE + (i < C::NE ? i : 0)];
        }
    };
    auto table_store = [&](uint32_t *E, const uint32_t(&e)[NEW]) {
#pragma unroll
      for (int j = 0; j < NEW; ++j)
        if (tid + j * NT < C::NE)
          E[tid + j * NT] = e[j];
    };
    // byte offset of the (first) DoF of a unit; constrained entity: out of range
    auto unit_offset = [&](uint32_t w, uint32_t m, int it = 0, uint32_t brick = 0) {
#ifdef MGX_MACRO_CONTIG // diagnostic build (wrong results): item i of brick b at address b NB^3 p^3 + i -- the access pattern of a contiguous stream
      return w != kInvalid ? ((brick_first + brick) * (uint32_t)(C::NB * C::NB * C::NB * P * P * P) + (uint32_t)(tid + it * NT)) * (uint32_t)sizeof(T) : kOob;
#else
      return w != kInvalid ? (ent_index(w) + item_offset(m)) * (uint32_t)sizeof(T) : kOob;
#endif
    };
    // read_dof_values_compressed through the entity table E: issues the loads of this thread's
    // items (constrained entity: out of range, reads zero -- vector_access_reduced.h:174-179)
    // kChebFirstProlong: the (PB p + 1)^3 coarse values of the brick's parents, loaded with the gather
    constexpr bool kProlong = MODE == kChebFirstProlong;
    constexpr int  CNP = (C::NB / 2) * P + 1, NCV = CNP * CNP * CNP / NT + 1, CE3 = (C::NB + 1) * (C::NB + 1) * (C::NB + 1);
    T              cvv[NCV];
    T    g[NG][IT];
    uint32_t cw[NCV]; // first coarse DoF of the entities of this thread's coarse points (next brick)
    // The coarse values travel ahead of the source: their table words are requested with the entity table of the
    // next brick (loop top, in flight during the sweeps), the values before the write-out (they land under it), and
    // the interpolation sweeps then run while the gathered source values are still in flight -- nothing of the
    // prolongation waits for memory after the first brick (phase stamps before: 21 500 cycles between the write-out and
    // the next loop top, as much as the write-out and the sweeps together; colour launch on one box, A/B: 188.7 -> 183 us)
    auto coarse_words = [&](uint32_t brick) {
      if (kProlong)
        {
          const uint32_t *ctab = post.coarse_blocks + (size_t)(brick_first + brick) * CE3;
#pragma unroll
          for (int k = 0; k < NCV; ++k)
            {
              const int l = tid + k * NT;
              int       slot = 0, pnt;
              uint32_t  off;
              if (l < CNP * CNP * CNP)
                coarse_point<P>(l, slot, off, pnt);
              cw[k] = ctab[slot];
            }
        }
    };
    auto coarse_values = [&]() {
      if (kProlong)
        {
#pragma unroll
          for (int k = 0; k < NCV; ++k)
            {
              const int l = tid + k * NT;
              cvv[k]      = T(0);
              if (l < CNP * CNP * CNP)
                {
                  int      slot, pnt;
                  uint32_t off;
                  coarse_point<P>(l, slot, off, pnt);
                  const uint32_t w = cw[k];
                  const T        v = post.coarse[w != kInvalid ? w + off : 0u];
                  cvv[k]           = w != kInvalid ? v : T(0); // constrained coarse DoF: zero
                }
            }
        }
    };
    auto gather_issue = [&](const uint32_t *E, uint32_t brick) {
      const rsrc_t r0 = MODE == kChebInit ? (DTAB ? R.a : R.b) : rsrc;
      const rsrc_t r1 = MODE == kCgUpdate ? R.b : R.a; // second operand
#pragma unroll
      for (int j = 0; j < JP; ++j)
        {
#pragma unroll
          for (int e = 0; e < 2; ++e)
            g[0][2 * j + e] = g[NG - 1][2 * j + e] = T(0);
          if (live_p(j))
            {
              const uint32_t off = unit_offset(E[item_slot(mw[2 * j])], mw[2 * j], 2 * j, brick);
              buf_ld2(r0, off, g[0][2 * j], g[0][2 * j + 1]);
              if (NG == 2)
                buf_ld2(r1, off, g[NG - 1][2 * j], g[NG - 1][2 * j + 1]);
            }
        }
#pragma unroll
      for (int j = 0; j < JS; ++j)
        {
          const int v = 2 * JP + j;
          g[0][v] = g[NG - 1][v] = T(0);
          if (live_s(j))
            {
              const uint32_t off = unit_offset(E[item_slot(mw[v])], mw[v], v, brick);
              g[0][v]            = buf_ld(r0, off, T());
              if (NG == 2)
                g[NG - 1][v] = buf_ld(r1, off, T());
            }
        }
    };
    T    xs[kKeepX ? IT : 1];
    auto gather_land = [&]() {
#pragma unroll
      for (int it = 0; it < IT; ++it)
        {
          T v = g[0][it];
          if (MODE == kChebInit) // x_1 = (1/theta) D^-1 b, never stored
            v = DTAB ? post.f0 * dv[DTAB ? it : 0] * g[0][it] : post.f0 * g[0][it] * g[NG - 1][it];
          if (MODE == kCgUpdate) // p = beta p + q, or p = q in the first step (laplace_operator.h:660-688)
            v = post.f1 == T(0) ? g[NG - 1][it] : post.f2 * g[0][it] + g[NG - 1][it];
          if (live(it))
            U[item_point(mw[it])] = v;
          if (kKeepX)
            xs[it] = v;
        }
    };

    // kChebFirstProlong: x += P x_coarse on the brick array and in the registers (W is free here)
    auto interpolate = [&]() {
      if (kProlong)
        {
          lds_barrier(); // the entity table parked in W (prologue) has been read by everyone
          prolong_brick<P, T, NT>(tid, W, B->P1eo, cvv);
        }
    };
    auto add_correction = [&]() {
      if (kProlong)
        {
#pragma unroll
          for (int it = 0; it < IT; ++it)
            if (live(it))
              {
                const uint32_t pnt = item_point(mw[it]);
                xs[kKeepX ? it : 0] += W[pnt];
                U[pnt] = xs[kKeepX ? it : 0];
              }
        }
    };

    // ---- prologue: table and source of the first brick ----
    table_load(b, ec);
    table_store(reinterpret_cast<uint32_t *>(W), ec);
    __syncthreads();
    MGX_STAMP(1);
    gather_issue(reinterpret_cast<const uint32_t *>(W), b);
    coarse_words(b);
    coarse_values();
    interpolate();
    gather_land();
    add_correction();
    MGX_STAMP(2);
    __syncthreads();

#ifdef MGX_MACRO_STAMPS
    int mgx_iter = 0;
#endif
    for (;;)
      {
        if (mgx_iter_is(4))
          MGX_STAMP(11);
        const uint32_t bn       = b + gridDim.x;
        const bool     has_next = bn < brick_count;
        MGX_STAMP_IT(3);
        if (has_next)
          {
            table_load(bn, en); // in flight during the sweeps
            coarse_words(bn);
          }
        // keep what is derived from the item words (LDS addresses, offsets) out of the registers
        // that live across the sweeps: the compiler must not hoist it out of the brick loop
#pragma unroll
        for (int it = 0; it < IT; ++it)
          asm volatile("" : "+v"(mw[it]));
        if (DTAB && (MODE == kChebInit || MODE == kChebOldInit)) // likewise f0 * dv
          {
#pragma unroll
            for (int it = 0; it < IT; ++it)
              asm volatile("" : "+v"(dv[DTAB ? it : 0]));
          }
#ifndef MGX_MACRO_NOSWEEP // diagnostic build without the sweeps (wrong results): memory phases alone
        if (kSlicedSweeps<P, MODE>)
          brick_sweeps_sliced<P, T>(
            tid, U, W, B, c0, c1, c2, [&](int i) { MGX_STAMP_IT(4 + i); }, [&]() { __syncthreads(); });
        else
          {
        // ---- x sweep: line l = (y,z), contiguous; M u -> W, K u -> U in place ----
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
        // the REM lines beyond the thread count, one cell block per thread (NB consecutive lanes per
        // line): the block's first output belongs to the previous cell's thread, which fetches it
        // from its neighbour lane.  In-place updates are safe: the lanes of a line sit in one wave.
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
        __syncthreads();
        MGX_STAMP_IT(4);
        // ---- y sweep: line l = (x,z), stride G ----
        if (REM > 0 && tid < REM * NB)
          {
            const int l = NT + tid / NB, c = tid % NB, base = (l / G) * (G * G) + l % G + c * P * G;
            T         seg[N], y[N], z[N], r[N];
#pragma unroll
            for (int i = 0; i < N; ++i)
              seg[i] = W[base + i * G];
            cell_apply2<P, T>(M, K, seg, y, z); // y: M t1, z: K t1
#pragma unroll
            for (int i = 0; i < N; ++i)
              seg[i] = U[base + i * G];
            cell_apply<P, T>(M, seg, r); // M k1
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
        __syncthreads();
        MGX_STAMP_IT(5);
        // ---- z sweep: line l = (x,y), stride G^2; result -> W ----
        if (REM > 0 && tid < REM * NB)
          {
            const int l = NT + tid / NB, c = tid % NB, base = l + c * P * (G * G);
            T         seg[N], y[N], r[N];
#pragma unroll
            for (int i = 0; i < N; ++i)
              seg[i] = W[base + i * (G * G)];
            cell_apply<P, T>(K, seg, r); // K t2
#pragma unroll
            for (int i = 0; i < N; ++i)
              seg[i] = U[base + i * (G * G)];
            cell_apply<P, T>(M, seg, y); // M s2
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
#endif
        __syncthreads();
        MGX_STAMP_IT(6);
        // ---- U is free: park both tables there and issue the next gather ----
        uint32_t *ebase = reinterpret_cast<uint32_t *>(U), *E2 = ebase + C::NE;
        table_store(ebase, ec);
        if (has_next)
          table_store(E2, en);
        const uint32_t *sbase = ebase + 2 * C::NE;
        if (FREE)
          {
#pragma unroll
            for (int j = 0; j < NEW; ++j)
              if (tid + j * NT < C::NE)
                ebase[2 * C::NE + tid + j * NT] = sc[FREE ? j : 0];
          }
        __syncthreads();
        // (kChebInit gathers two operands per value: too many registers in flight next to the
        // write-out, its gather is issued afterwards)
        // (likewise kChebFirstProlong, which also holds the coarse values and the corrected x: with the next
        // gather in flight it spills 56 B per lane; without, 177 instead of 193 us per colour launch)
        constexpr bool kPipeGather = (MODE != kChebInit || DTAB) && MODE != kCgUpdate && MODE != kChebFirstProlong;
        if (has_next && kPipeGather)
          gather_issue(E2, bn);
        if (has_next)
          coarse_values(); // land under the write-out
        MGX_STAMP_IT(7);

        // ---- write-out with the fused post-operation, same item -> thread mapping as the gather.
        // Units (pairs, then singles) in chunks, software-pipelined: the loads of chunk c + 1 are
        // issued before chunk c is computed and stored, so no load ever has to wait for an older
        // store (vmcnt counts loads and stores in issue order) and two chunks of operands are in
        // flight.  Both DoFs of a pair belong to one entity: same flags, same target vector. ----
        {
          // units per chunk: everything at once where few operands are loaded (plain), fewer where
          // the fused Chebyshev forms keep three or four operand pairs per unit in flight
          constexpr int kChunk = MODE == kPlain ? NU : (MODE == kResidual ? (NU + 1) / 2 : (kKeepX ? MGX_MACRO_CHUNK_CHEB : MGX_MACRO_CHUNK)),
                        NCH    = (NU + kChunk - 1) / kChunk;
          PostOps<T>    ops[2][kChunk];
          uint32_t      iw[2][kChunk]; // entity table word of the unit (kInvalid: nothing to do)
          uint32_t      pw[2][FREE ? kChunk : 1]; // FREE: byte offset of the unit's private value (interior: out of range)
          const uint32_t priv0 = FREE ? (brick_first + b) * post.n_surf : 0u;
          // unit u: pair u (u < JP) or single u - JP; v0 = its first value slot
          auto slot0 = [&](int u) { return u < JP ? 2 * u : JP + u; };
          auto ulive = [&](int u) { return u < JP ? live_p(u) : live_s(u - JP); };
          auto issue = [&](int c, PostOps<T>(&o)[kChunk], uint32_t(&w)[kChunk], uint32_t(&pv)[FREE ? kChunk : 1]) {
            // table lookups of the whole chunk first (one batch of LDS reads), then the loads
#pragma unroll
            for (int j = 0; j < kChunk; ++j)
              {
                const int u = c * kChunk + j;
                w[j]        = kInvalid;
                if (u < NU && ulive(u))
                  w[j] = ebase[item_slot(mw[slot0(u)])];
                if (FREE)
                  {
                    // a private entity: FIRST (nothing to add to), never LAST, its value goes to the
                    // brick's private block; every other entity follows the flags of its table word
                    const uint32_t so = (u < NU && ulive(u)) ? sbase[item_slot(mw[slot0(u < NU ? u : 0)])] : kInvalid;
                    const bool     vd = w[j] != kInvalid;
                    if (vd && so != kInvalid)
                      w[j] = ent_index(w[j]) | 0x40000000u;
                    pv[FREE ? j : 0] = (vd && so != kInvalid)
                                         ? (priv0 + so + item_offset(mw[slot0(u < NU ? u : 0)])) * (uint32_t)sizeof(T)
                                         : kOob;
                  }
              }
#pragma unroll
            for (int j = 0; j < kChunk; ++j)
              {
                const int u = c * kChunk + j;
                if (u < JP)
                  post_issue<T, MODE, DTAB, true>(R, w[j], unit_offset(w[j], mw[slot0(u < NU ? u : 0)], slot0(u < NU ? u : 0), b), o[j]);
                else if (u < NU)
                  post_issue<T, MODE, DTAB, false>(R, w[j], unit_offset(w[j], mw[slot0(u < NU ? u : 0)], slot0(u < NU ? u : 0), b), o[j]);
              }
          };
          issue(0, ops[0], iw[0], pw[0]);
#pragma unroll
          for (int c = 0; c < NCH; ++c)
            {
              if (c + 1 < NCH)
                issue(c + 1, ops[(c + 1) & 1], iw[(c + 1) & 1], pw[(c + 1) & 1]);
              __builtin_amdgcn_sched_barrier(0);
              T val[kChunk][2];
#pragma unroll
              for (int j = 0; j < kChunk; ++j)
                {
                  const int u = c * kChunk + j;
                  val[j][0] = val[j][1] = T(0);
                  if (u < NU && ulive(u))
                    {
                      val[j][0] = W[item_point(mw[slot0(u)])];
                      if (u < JP)
                        val[j][1] = W[item_point(mw[slot0(u) + 1])];
                    }
                }
#pragma unroll
              for (int j = 0; j < kChunk; ++j)
                {
                  const int u = c * kChunk + j;
                  if (u < NU)
                    {
                      const int         v0   = slot0(u);
                      const bool        pair = u < JP;
                      const uint32_t    w    = iw[c & 1][j];
                      const bool        vld  = w != kInvalid, last = vld && (w >> 31);
                      const uint32_t    off  = unit_offset(w, mw[v0], v0, b);
                      const PostOps<T> &o    = ops[c & 1][j];
                      T                 res[2];
#pragma unroll
                      for (int e = 0; e < 2; ++e)
                        res[e] = (e == 0 || pair)
                                   ? post_finish<T, MODE>(post, o.pv[e], o.av[e], o.ov[e], DTAB ? dv[DTAB ? v0 + e : 0] : o.bv[e],
                                                          last, val[j][e], xs[kKeepX ? v0 + e : 0])
                                   : T(0);
                      constexpr int kStAux = (MODE == kPlain || MODE == kResidual) ? kAuxNt : 0;
                      auto          st     = [&](rsrc_t r, uint32_t f) {
                        if (pair)
                          buf_st2(r, f, res[0], res[1]);
                        else
                          buf_st<kStAux>(r, f, res[0]);
                      };
                      if (MODE == kResidualRestrict)
                        {
                          // the brick's share of the residual stays in W for the restriction (rows of constrained
                          // DoFs: zero)
                          if (ulive(u))
                            {
                              W[item_point(mw[v0])] = vld ? res[0] : T(0);
                              if (pair)
                                W[item_point(mw[v0 + 1])] = vld ? res[1] : T(0);
                            }
                        }
                      else if (MODE == kCgUpdate)
                        {
                          // completion: q = A p, p = p_new, x += alpha p_old and the four sums;
                          // otherwise the partial sum goes to the carrier (q still holds z there)
                          st(R.out, last ? off : kOob);
                          if (__builtin_amdgcn_ballot_w64(vld && !last) != 0)
                            st(R.partial, last ? kOob : off);
                          const T xn = o.ov[0] + post.f1 * o.bv[0];
                          buf_st(R.srcw, last ? off : kOob, xs[kKeepX ? v0 : 0]);
                          if (post.f1 != T(0))
                            buf_st(R.xw, last ? off : kOob, xn);
                          if (last)
                            {
                              const double qv = (double)res[0], pn = (double)xs[kKeepX ? v0 : 0], rv = (double)o.av[0];
                              cg[0] += qv * pn;
                              cg[1] += rv * rv;
                              cg[2] += qv * rv;
                              cg[3] += qv * qv;
                            }
                        }
                      else
                        {
                          // out-of-range offsets drop the store; whole waves of interior items skip
                          // the carrier store
                          st(R.out, last ? off : kOob);
                          if (FREE)
                            {
                              const uint32_t po = pw[c & 1][FREE ? j : 0];
                              if (__builtin_amdgcn_ballot_w64(po != kOob) != 0)
                                st(R.priv, po);
                              if (__builtin_amdgcn_ballot_w64(vld && !last && po == kOob) != 0)
                                st(R.partial, (last || po != kOob) ? kOob : off);
                            }
#ifndef MGX_MACRO_NOCARRIER
                          else if (__builtin_amdgcn_ballot_w64(vld && !last) != 0)
                            st(R.partial, last ? kOob : off);
#endif
                          if (MODE == kChebFirstProlong) // the corrected x is x_old of the next iteration
                            buf_st(R.srcw, last ? off : kOob, xs[kKeepX ? v0 : 0]);
                        }
                    }
                }
              __builtin_amdgcn_sched_barrier(0);
            }
        }
        MGX_STAMP_IT(8);
        if (MODE == kResidualRestrict)
          {
            constexpr int CE1 = C::NB + 1; // 2 PB + 1
            __syncthreads();
            const uint32_t *ctab = post.coarse_blocks + (size_t)(brick_first + b) * (CE1 * CE1 * CE1);
            if (post.coarse_scratch) // uniform
              restrict_brick<P, T, NT, true>(tid, W, B->P1eo, nullptr, ctab,
                                             post.coarse_scratch + (size_t)(brick_first + b) * (CNP * CNP * CNP));
            else
              restrict_brick<P, T, NT, false>(tid, W, B->P1eo, post.coarse, ctab);
          }
        if (!has_next)
          break;
        if (!kPipeGather)
          gather_issue(E2, bn);
        __syncthreads(); // everyone is done with the parked tables and with W
#pragma unroll
        for (int j = 0; j < NEW; ++j)
          ec[j] = en[j];
        interpolate(); // (kChebFirstProlong) on W, while the source values are in flight
        gather_land();
        add_correction();
        MGX_STAMP_IT(9);
        __syncthreads();
        MGX_STAMP_IT(10);
        b = bn;
        MGX_STAMP_NEXT();
      }
    if (MODE == kCgUpdate)
      {
        // deterministic tree reduction of the four sums over the workgroup (U is free)
        static_assert(MODE != kCgUpdate || !MGX_MACRO_PAIRS, "kCgUpdate is written for single items");
        __syncthreads();
        double *red = reinterpret_cast<double *>(U);
#pragma unroll
        for (int k = 0; k < 4; ++k)
          red[k * NT + tid] = cg[k];
        __syncthreads();
        constexpr int P2 = NT <= 64 ? 64 : (NT <= 128 ? 128 : (NT <= 256 ? 256 : 512));
        for (int stride = P2 / 2; stride > 0; stride >>= 1)
          {
            if (tid < stride && tid + stride < NT)
              {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                  red[k * NT + tid] += red[k * NT + tid + stride];
              }
            __syncthreads();
          }
        if (tid < 4)
          post.sums[blockIdx.x * 4 + tid] = red[tid * NT];
      }
#ifdef MGX_MACRO_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MGX_STAMP(14);
    MGX_STAMP(13);
#endif
  }

  // ------------------------------------------------------------------------------------------
  // workgroups of a persistent launch per resident slot: the CUs of the device (x WGS per CU);
  // Tunables::macro_wg_x16 scales it (tuning aid)
  static uint32_t macro_cus(const OperatorData &op)
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
  static void macro_launch_free(hipStream_t s, const OperatorData &op, const T *src, const BrickPost<T> &post_in, int g0, int g1)
  {
    using C             = MCfg<P, T>;
    const FreeSchedule &fr = op.bricks.fr;
    BrickPost<T>        post = post_in;
    post.priv       = (T *)fr.priv;
    post.surf_off   = fr.surf_off;
    post.n_surf     = fr.n_surf;
    post.priv_bytes = (uint32_t)((size_t)op.bricks.n_bricks * fr.n_surf * sizeof(T));
    constexpr bool kUsesDiag = is_cheb_mode(MODE);
    if (kUsesDiag && op.diag_items)
      post.b = (const T *)op.diag_items;
    for (int g = g0; g < g1; ++g)
      {
        const uint32_t first = fr.group_start[g], count = fr.group_start[g + 1] - first;
        if (count == 0)
          continue;
        const uint32_t grid = std::min<uint32_t>(count, (uint32_t)(op.macro_wg_x16 ? 1 : C::WGS) * macro_cus(op));
        if (kUsesDiag && op.diag_items)
          hipLaunchKernelGGL((brick_macro_kernel<P, T, MODE, kUsesDiag, true>), dim3(grid), dim3(C::THREADS), 0, s, src, first,
                             count, fr.ent, op.bricks.item_map, (const Basis1D<T> *)op.basis, (T)op.coef[0], (T)op.coef[1],
                             (T)op.coef[2], post, (uint32_t)(op.n_dofs * sizeof(T)));
        else
          hipLaunchKernelGGL((brick_macro_kernel<P, T, MODE, false, true>), dim3(grid), dim3(C::THREADS), 0, s, src, first,
                             count, fr.ent, op.bricks.item_map, (const Basis1D<T> *)op.basis, (T)op.coef[0], (T)op.coef[1],
                             (T)op.coef[2], post, (uint32_t)(op.n_dofs * sizeof(T)));
      }
  }

  template <int P, typename T, int MODE>
  static void macro_launch(hipStream_t s, const OperatorData &op, const T *src, const BrickPost<T> &post, int g0, int g1,
                           bool free_schedule = false)
  {
    using C             = MCfg<P, T>;
    const BrickData &bd = op.bricks;
    if (free_schedule)
      {
        if constexpr (MODE <= kChebOldInit)
          return macro_launch_free<P, T, MODE>(s, op, src, post, g0, g1);
      }
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
        // persistent workgroups: as many as are resident at once (WGS per CU), each walks over
        // count / grid bricks
        const uint32_t grid = std::min<uint32_t>(count, (uint32_t)(op.macro_wg_x16 ? 1 : C::WGS) * macro_cus(op));
        constexpr bool kUsesDiag = is_cheb_mode(MODE);
        if (kUsesDiag && op.diag_items)
          {
            BrickPost<T> pt = post;
            pt.b            = (const T *)op.diag_items;
            hipLaunchKernelGGL((brick_macro_kernel<P, T, MODE, kUsesDiag>), dim3(grid), dim3(C::THREADS), 0, s, src, first,
                               count, bd.ent_base, bd.item_map, (const Basis1D<T> *)op.basis, (T)op.coef[0],
                               (T)op.coef[1], (T)op.coef[2], pt, (uint32_t)(op.n_dofs * sizeof(T)));
          }
        else
          hipLaunchKernelGGL((brick_macro_kernel<P, T, MODE, false>), dim3(grid), dim3(C::THREADS), 0, s, src, first,
                             count, bd.ent_base, bd.item_map, (const Basis1D<T> *)op.basis, (T)op.coef[0],
                             (T)op.coef[1], (T)op.coef[2], post, (uint32_t)(op.n_dofs * sizeof(T)));
      }
  }

  template <int P, typename T>
  static void macro_modes(hipStream_t s, const OperatorData &op, int mode, const T *src, const BrickPost<T> &post, int g0,
                          int g1, bool fr)
  {
    switch (mode)
      {
        case kPlain: macro_launch<P, T, kPlain>(s, op, src, post, g0, g1, fr); break;
        case kResidual: macro_launch<P, T, kResidual>(s, op, src, post, g0, g1, fr); break;
        case kCheb: macro_launch<P, T, kCheb>(s, op, src, post, g0, g1, fr); break;
        case kChebFirst: macro_launch<P, T, kChebFirst>(s, op, src, post, g0, g1, fr); break;
        case kChebZeroOld: macro_launch<P, T, kChebZeroOld>(s, op, src, post, g0, g1, fr); break;
        case kChebInit: macro_launch<P, T, kChebInit>(s, op, src, post, g0, g1, fr); break;
        case kChebOldInit: macro_launch<P, T, kChebOldInit>(s, op, src, post, g0, g1, fr); break;
        case kResidualRestrict: macro_launch<P, T, kResidualRestrict>(s, op, src, post, g0, g1); break;
        case kChebFirstProlong: macro_launch<P, T, kChebFirstProlong>(s, op, src, post, g0, g1); break;
        default: break;
      }
  }

  // ------------------------------------------------------------------------------------------
#if MGX_MACRO_IS_F64
  // sums[k] = extra[k] + sum of the n partial quadruples, in index order (deterministic)
  __global__ void reduce4_kernel(const double *__restrict__ partials, uint32_t n, const double *__restrict__ extra,
                                 double *__restrict__ sums)
  {
    __shared__ double red[4 * 256];
    const int         tid = threadIdx.x;
    double            acc[4] = {0., 0., 0., 0.};
    for (uint32_t i = tid; i < n; i += 256)
      for (int k = 0; k < 4; ++k)
        acc[k] += partials[4 * (size_t)i + k];
    for (int k = 0; k < 4; ++k)
      red[k * 256 + tid] = acc[k];
    __syncthreads();
    for (int stride = 128; stride > 0; stride >>= 1)
      {
        if (tid < stride)
          for (int k = 0; k < 4; ++k)
            red[k * 256 + tid] += red[k * 256 + tid + stride];
        __syncthreads();
      }
    if (tid < 4)
      sums[tid] = red[tid * 256] + (extra ? extra[tid] : 0.);
  }
#endif

  // ------------------------------------------------------------------------------------------
  // Item-ordered table of the inverse diagonal (DTAB).  collect: every brick writes the value of
  // each of its unconstrained items into the table (all bricks write the same value if the diagonal
  // is periodic); verify: any item of any brick that differs bitwise from the table raises the flag.
  template <typename T, bool VERIFY>
  __global__ void diag_table_kernel(const T *__restrict__ inv_diag, const uint32_t *__restrict__ ent_base,
                                    const uint32_t *__restrict__ item_map, uint32_t ne, uint32_t npts, T *table,
                                    uint32_t *flag)
  {
    const uint32_t brick = blockIdx.x;
    for (uint32_t i = threadIdx.x; i < npts; i += blockDim.x)
      {
        const uint32_t m = item_map[i], w = ent_base[(size_t)brick * ne + item_slot(m)];
        if (w == kInvalid)
          continue;
        const T v = inv_diag[ent_index(w) + item_offset(m)];
        if (!VERIFY)
          table[i] = v;
        else
          {
            // equal up to the rounding of the assembly: the cell contributions to one diagonal entry
            // are added in the order the atomics arrive, which differs from brick to brick in the
            // last bits (a few ulp); anything beyond that means the diagonal is not periodic
            const T    t   = table[i];
            const T    tol = (sizeof(T) == 8 ? T(1e-14) : T(2e-6)) * (t < 0 ? -t : t);
            const T    dvt = v - t;
            const bool same = (dvt < 0 ? -dvt : dvt) <= tol;
            if (!same)
              atomicOr(flag, 1u);
          }
      }
  }

#define MGX_CAT2(a, b) a##b
#define MGX_CAT(a, b) MGX_CAT2(a, b)
  // builds the table into `table` (device, (NB p + 1)^3 values, zero-initialised by the caller);
  // *flag_dev (device, zero-initialised) is nonzero afterwards if the diagonal is not periodic
  void MGX_CAT(macro_diag_table_, MGX_MACRO_SUFFIX)(hipStream_t s, const OperatorData &op, const uint32_t *item_map, void *table,
                                                    uint32_t *flag_dev)
  {
    using T             = MGX_MACRO_T;
    const BrickData &bd = op.bricks;
    const uint32_t   nb = op.p <= 4 ? 4 : 2, g = nb * op.p + 1, npts = g * g * g, e1 = 2 * nb + 1, ne = e1 * e1 * e1;
    hipLaunchKernelGGL((diag_table_kernel<T, false>), dim3(bd.n_bricks), dim3(256), 0, s, (const T *)op.inv_diag,
                       bd.ent_base, item_map, ne, npts, (T *)table, flag_dev);
    hipLaunchKernelGGL((diag_table_kernel<T, true>), dim3(bd.n_bricks), dim3(256), 0, s, (const T *)op.inv_diag,
                       bd.ent_base, item_map, ne, npts, (T *)table, flag_dev);
  }

  // vmult_with_cg_update on a brick-scheduled level of one rank (mgx_api.cpp handles the rest):
  // partials receives gridDim x 4 doubles per launch group, *n_partials their total count
  bool MGX_CAT(launch_macro_cg_update_, MGX_MACRO_SUFFIX)(hipStream_t s, const OperatorData &op, double alpha, double beta,
                                                           const void *r, void *q, void *p, void *x, void *carrier,
                                                           double *partials, uint32_t capacity, uint32_t *n_partials)
  {
    using T = MGX_MACRO_T;
    if (MGX_MACRO_PAIRS || (uint64_t)op.n_dofs * sizeof(T) >= 0xFFFFFFF0ull)
      return false;
    BrickPost<T> post{};
    post.a       = (const T *)r;
    post.b       = (const T *)q;
    post.old     = (const T *)x;
    post.out     = (T *)q;
    post.partial = (T *)carrier;
    post.f1      = (T)alpha;
    post.f2      = (T)beta;
    post.src_w   = (T *)p;
    post.x_w     = (T *)x;
    const BrickData &bd   = op.bricks;
    uint32_t         used = 0;
    auto             run  = [&](auto cfg) {
      using C = decltype(cfg);
      // room for the partial sums of ALL launch groups is checked before the first launch: a launch
      // already updates p, q and x, after which the caller's unfused path would start from half-updated vectors
      uint64_t total = 0;
      for (int c = 0; c < bd.n_colours; ++c)
        total += std::min<uint32_t>(bd.colour_start[c + 1] - bd.colour_start[c],
                                    (uint32_t)(op.macro_wg_x16 ? 1 : C::WGS) * macro_cus(op));
      if (total > capacity)
        return false;
      for (int c = 0; c < bd.n_colours; ++c)
        {
          const uint32_t first = bd.colour_start[c], count = bd.colour_start[c + 1] - first;
          if (count == 0)
            continue;
          const uint32_t grid = std::min<uint32_t>(count, (uint32_t)(op.macro_wg_x16 ? 1 : C::WGS) * macro_cus(op));
          post.sums = partials + 4 * (size_t)used;
          used += grid;
          hipLaunchKernelGGL((brick_macro_kernel<C::N - 1, T, kCgUpdate, false>), dim3(grid), dim3(C::THREADS), 0, s,
                             (const T *)p, first, count, bd.ent_base, bd.item_map, (const Basis1D<T> *)op.basis,
                             (T)op.coef[0], (T)op.coef[1], (T)op.coef[2], post, (uint32_t)(op.n_dofs * sizeof(T)));
        }
      return true;
    };
    bool ok = false;
    switch (op.p)
      {
#ifdef MGX_MACRO_ONLY_P
        case MGX_MACRO_ONLY_P: ok = run(MCfg<MGX_MACRO_ONLY_P, T>()); break;
#else
        case 1: ok = run(MCfg<1, T>()); break;
        case 2: ok = run(MCfg<2, T>()); break;
        case 3: ok = run(MCfg<3, T>()); break;
        case 4: ok = run(MCfg<4, T>()); break;
        case 5: ok = run(MCfg<5, T>()); break;
        case 6: ok = run(MCfg<6, T>()); break;
        case 7: ok = run(MCfg<7, T>()); break;
        case 8: ok = run(MCfg<8, T>()); break;
        case 9: ok = run(MCfg<9, T>()); break;
#endif
        default: break;
      }
    *n_partials = used;
    return ok;
  }

#if MGX_MACRO_IS_F64
  void launch_reduce4(hipStream_t s, const double *partials, uint32_t n, const double *extra, double *sums)
  {
    hipLaunchKernelGGL(reduce4_kernel, dim3(1), dim3(256), 0, s, partials, n, extra, sums);
  }
#endif

  // one translation unit per number type (Makefile: -DMGX_MACRO_T=double|float -DMGX_MACRO_SUFFIX=f64|f32)
  bool MGX_CAT(launch_macro_loop_, MGX_MACRO_SUFFIX)(hipStream_t s, const OperatorData &op, int mode, const void *src,
                                                     const void *a, const void *b, void *out, void *partial,
                                                     double f1, double f2, const void *old, double f0, void *coarse,
                                                     const uint32_t *coarse_blocks, int g0, int g1, bool free_schedule)
  {
    using T = MGX_MACRO_T;
    const bool fr = free_schedule && op.bricks.fr.available() && mode <= kChebOldInit && !MGX_MACRO_PAIRS;
    if (free_schedule && !fr)
      return false;
    // the forms the second pipeline covers (mgx_macro2.hip) on the eight-colour schedule
    if (!free_schedule && op.macro_v2 && !MGX_MACRO_PAIRS &&
        MGX_CAT(launch_macro2_loop_, MGX_MACRO_SUFFIX)(s, op, mode, src, a, out, partial, coarse, coarse_blocks, g0, g1, f1, f2, f0, old))
      return true;
    if (mode < kPlain || (mode > kResidualRestrict && mode != kChebFirstProlong) || MGX_MACRO_PAIRS * (mode == kChebFirstProlong) ||
        (uint64_t)op.n_dofs * sizeof(T) >= 0xFFFFFFF0ull)
      return false;
    BrickPost<T> post{};
    post.a             = (const T *)a;
    post.b             = (const T *)b;
    post.old           = (const T *)old;
    post.out           = (T *)out;
    post.partial       = (T *)partial;
    post.f1            = (T)f1;
    post.f2            = (T)f2;
    post.f0            = (T)f0;
    post.coarse        = (T *)coarse;
    post.coarse_blocks = coarse_blocks;
    // (kResidualRestrict carries no partial sums: its `partial` argument names the coarse scratch array, if any)
    post.coarse_scratch = mode == kResidualRestrict ? (T *)partial : nullptr;
    post.src_w         = (T *)const_cast<void *>(src); // kChebFirstProlong writes the corrected x back
    switch (op.p)
      {
#ifdef MGX_MACRO_ONLY_P
        case MGX_MACRO_ONLY_P: macro_modes<MGX_MACRO_ONLY_P, T>(s, op, mode, (const T *)src, post, g0, g1, fr); break;
#else
        case 1: macro_modes<1, T>(s, op, mode, (const T *)src, post, g0, g1, fr); break;
        case 2: macro_modes<2, T>(s, op, mode, (const T *)src, post, g0, g1, fr); break;
        case 3: macro_modes<3, T>(s, op, mode, (const T *)src, post, g0, g1, fr); break;
        case 4: macro_modes<4, T>(s, op, mode, (const T *)src, post, g0, g1, fr); break;
        case 5: macro_modes<5, T>(s, op, mode, (const T *)src, post, g0, g1, fr); break;
        case 6: macro_modes<6, T>(s, op, mode, (const T *)src, post, g0, g1, fr); break;
        case 7: macro_modes<7, T>(s, op, mode, (const T *)src, post, g0, g1, fr); break;
        case 8: macro_modes<8, T>(s, op, mode, (const T *)src, post, g0, g1, fr); break;
        case 9: macro_modes<9, T>(s, op, mode, (const T *)src, post, g0, g1, fr); break;
#endif
        default: return false;
      }
    return true;
  }
} // namespace mgx
