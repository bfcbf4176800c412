// mgx_transfer.hip -- level transfers of the V-cycle (MGTransferMatrixFree restated, SURVEY.md 8a
// row R; multigrid_solver.h:415, 668, 674), software-pipelined.
//
// Work decomposition
//   * a parent cell and its 8 children form a patch of (2p+1)^3 fine points = 5^3 mesh entities
//     (vertices, lines, quads, hexes of the children).  The host precomputes one table row of 125
//     words per parent (mgx_api.cpp, build_patch_table): first fine DoF of the entity, the weight
//     1/multiplicity as a shift (restriction) and an ownership bit (prolongation: the fine DoFs of
//     an entity are written by exactly one parent).  One coalesced 500-B read replaces the chain
//     children -> 8 rows of the fine index table -> data of the first version (mgx_kernels.hip).
//   * workgroups are persistent: each walks over parents pc = blockIdx.x + k gridDim.x and keeps a
//     three-stage pipeline in flight -- table row of parent k+2, vector data of parent k+1,
//     sum-factorised interpolation of parent k -- so that no stage waits for an HBM round trip.
//     Both kernels only move data (8 B/DoF read for the restriction, 16 B/DoF for the adding
//     prolongation); without the pipeline they were bound by four dependent memory latencies per
//     workgroup rather than by bandwidth.
//   * every fine point of the patch is loaded once ((2p+1)^3 loads per parent instead of the
//     8 (p+1)^3 of a child-by-child gather).
//   * the three 1D sweeps run line-per-thread: a thread reads the 2p+1 (restriction) or p+1
//     (prolongation) values of one line once, keeps them in registers and forms all outputs of the
//     line with the entries of P1 as wave-uniform scalar operands.  (Forming one output per thread
//     with both operands read from LDS made the kernels LDS-bandwidth bound: 2 LDS reads per FMA.)
//     The z sweep works directly on the registers the patch values were prefetched into (restriction)
//     or are stored from (prolongation); only the two intermediate arrays live in LDS.
#include "mgx_brick_device.hpp" // restrict_half / prolong_half
#include "mgx_internal.hpp"

#include <hip/hip_runtime.h>

#include <cstdlib>
#include <mutex>

namespace mgx
{
  template <int P>
  struct TPCfg
  {
    static constexpr int N       = P + 1;
    static constexpr int M       = 2 * P + 1;
    static constexpr int THREADS = P <= 2 ? 64 : (P <= 4 ? 128 : 256);
    static constexpr int NL      = (M * M + THREADS - 1) / THREADS; // z-lines of the patch per thread
  };

  // patch table word: bits 0..28 first DoF, bits 29..30 log2(multiplicity), bit 31 owned
  __device__ __forceinline__ uint32_t pw_index(uint32_t w) { return w & 0x1FFFFFFFu; }
  __device__ __forceinline__ uint32_t pw_shift(uint32_t w) { return (w >> 29) & 3u; }
  __device__ __forceinline__ bool     pw_owned(uint32_t w) { return (w >> 31) != 0u; }

  // position of patch point a in [0, 2p] along one direction: entity layer (0..4), offset inside
  // the layer and the layer's size
  template <int P>
  __device__ __forceinline__ void patch_layer(int a, int &layer, int &off, int &size)
  {
    layer = a == 0 ? 0 : (a < P ? 1 : (a == P ? 2 : (a < 2 * P ? 3 : 4)));
    off   = layer == 1 ? a - 1 : (layer == 3 ? a - P - 1 : 0);
    size  = (layer & 1) ? P - 1 : 1;
  }

  // the z-line (a, b) of the patch: slot and offset of its point c are
  // slot = 25 ez(c) + sxy, offset = oz(c) nxy + oxy
  struct PatchLine
  {
    int sxy, oxy, nxy;
  };
  template <int P>
  __device__ __forceinline__ PatchLine patch_line(int a, int b)
  {
    int ex, ey, ox, oy, nx, ny;
    patch_layer<P>(a, ex, ox, nx);
    patch_layer<P>(b, ey, oy, ny);
    return {ey * 5 + ex, oy * nx + ox, ny * nx};
  }

  // x-line (j, k) of a coarse cell through its 27-entry row: entities 3 (3 cz + cy) + {0,1,2}
  template <int P>
  __device__ __forceinline__ void coarse_line(int j, int k, int &row, uint32_t &off)
  {
    const int cy = j == 0 ? 0 : (j == P ? 2 : 1), cz = k == 0 ? 0 : (k == P ? 2 : 1);
    const int oy = cy == 1 ? j - 1 : 0, oz = cz == 1 ? k - 1 : 0;
    row = 3 * (3 * cz + cy);
    off = (uint32_t)((cy == 1 ? P - 1 : 1) * oz + oy);
  }

  // persistent-workgroup plumbing shared by both kernels: double-buffered table rows in LDS
  template <int TH>
  struct TableStage
  {
    uint32_t wreg = 0, wreg2 = 0, creg = kInvalid;
    __device__ __forceinline__ void fetch(const uint32_t *__restrict__ patch, const uint32_t *__restrict__ idx_c,
                                          uint32_t parent, uint32_t n_parents, int tid, uint32_t stride = 1u)
    {
      const bool   ok  = parent < n_parents;
      const size_t row = (size_t)parent * stride;
      wreg             = (ok && tid < 125) ? patch[125u * row + tid] : 0u;
      if (TH < 125)
        wreg2 = (ok && tid + 64 < 125) ? patch[125u * row + tid + 64] : 0u;
      creg = (ok && tid < 27) ? idx_c[27u * row + tid] : kInvalid;
    }
    __device__ __forceinline__ void publish(uint32_t *tbl, uint32_t *ctb, int tid) const
    {
      if (tid < 125)
        tbl[tid] = wreg;
      if (TH < 125 && tid + 64 < 125)
        tbl[tid + 64] = wreg2;
      if (tid < 27)
        ctb[tid] = creg;
    }
  };

  // ------------------------------------------------------------------------------------------
  // restrict_and_add: coarse += P^T (w .* fine), w = 1/multiplicity of the fine DoF among the
  // parent patches.  Shared coarse DoFs receive the parents' contributions by atomic adds.
  // ------------------------------------------------------------------------------------------
  // COLOURED: the launch covers the parents 8 m + colour only.  Parents of one colour share no
  // coarse DoF (the host has verified it), so their sums are added with plain read-modify-writes
  // whose reads are prefetched with the patch data -- deterministic, and not limited by the
  // memory-side atomic units (scattered fp64 atomics run at a few % of the store rate).
  template <int P, typename T, bool COLOURED>
  __global__ void __launch_bounds__(TPCfg<P>::THREADS)
    restrict_pipe_kernel(T *__restrict__ coarse, const T *__restrict__ fine, const uint32_t *__restrict__ patch_,
                         const uint32_t *__restrict__ idx_c_, uint32_t n_parents_, const Basis1D<T> *__restrict__ B,
                         uint32_t colour, T *__restrict__ scratch, int owner_weights)
  {
    // scratch (uncoloured launch only): the parent's (p+1)^3 sums go to scratch[parent (p+1)^3 + .] for the
    // ordered assembly of the coarse level (mgx_kernels.hip, assemble_kernel) instead of atomic adds
    // a coloured launch sees the table rows of its parents as a strided array
    const uint32_t  n_parents = COLOURED ? n_parents_ / 8u : n_parents_;
    const uint32_t  pstride   = COLOURED ? 8u : 1u;
    const uint32_t *patch = patch_ + (COLOURED ? 125u * (size_t)colour : 0u);
    const uint32_t *idx_c = idx_c_ + (COLOURED ? 27u * (size_t)colour : 0u);
    using C          = TPCfg<P>;
    constexpr int N = C::N, M = C::M, TH = C::THREADS, NL = C::NL;
    __shared__ uint32_t tbl[2][128];
    __shared__ uint32_t ctb[2][32];
    __shared__ T        t2[N * M * M]; // [k][b][a]
    __shared__ T        t1[N * N * M]; // [k][j][a]
    const int      tid = threadIdx.x;
    const uint32_t G   = gridDim.x;
    const T       *pe  = B->P1eo; // the embedding in even-odd form (mgx_brick_device.hpp restrict_half), wave-uniform

    PatchLine line[NL];
    bool      has[NL];
#pragma unroll
    for (int it = 0; it < NL; ++it)
      {
        const int l = tid + it * TH;
        has[it]     = l < M * M;
        line[it]    = patch_line<P>(has[it] ? l % M : 0, has[it] ? l / M : 0);
      }
    // layer / offset of point c along z: compile-time after unrolling
    auto zslot = [](int c) {
      int e, o, n;
      patch_layer<P>(c, e, o, n);
      return e;
    };
    auto zoff = [](int c) {
      int e, o, n;
      patch_layer<P>(c, e, o, n);
      return o;
    };

    static_assert(N * N <= TH, "one coarse x-line per thread");
    const bool cl   = tid < N * N;
    int        crow = 0;
    uint32_t   coff = 0;
    coarse_line<P>(cl ? tid % N : 0, cl ? tid / N : 0, crow, coff);

    uint32_t        pc = blockIdx.x;
    TableStage<TH>  ts;
    T               v[NL][M];
    T               co[COLOURED ? N : 1]; // coloured: current coarse values of this thread's x-line
    uint32_t        sh[NL][5]; // shift of the five z-layers of the line
    auto            issue = [&](int b) {
      if (COLOURED)
        {
          const uint32_t b0 = ctb[b][crow], b1 = ctb[b][crow + 1], b2 = ctb[b][crow + 2];
          co[0]             = coarse[(cl && b0 != kInvalid) ? b0 + coff : 0u];
#pragma unroll
          for (int i = 1; i < P; ++i)
            co[COLOURED ? i : 0] = coarse[(cl && b1 != kInvalid) ? b1 + coff * (uint32_t)(P - 1) + (uint32_t)(i - 1) : 0u];
          co[COLOURED ? P : 0] = coarse[(cl && b2 != kInvalid) ? b2 + coff : 0u];
        }
#pragma unroll
      for (int it = 0; it < NL; ++it)
        {
          uint32_t w[5];
#pragma unroll
          for (int e = 0; e < 5; ++e)
            {
              w[e]      = tbl[b][has[it] ? 25 * e + line[it].sxy : 0];
              // weight 2^-shift of the fine DoFs of this patch entity; owner weights (multi-block meshes): the field
              // is 0 for the one parent (of the one rank) that restricts the entity -- weight 1 -- and 1 for the
              // others -- weight 0 (shift 31 is never a real one)
              sh[it][e] = owner_weights ? (pw_shift(w[e]) == 0u ? 0u : 31u) : pw_shift(w[e]);
            }
#pragma unroll
          for (int c = 0; c < M; ++c)
            v[it][c] = fine[has[it] ? pw_index(w[zslot(c)]) + (uint32_t)(zoff(c) * line[it].nxy + line[it].oxy) : 0u];
        }
    };

    ts.fetch(patch, idx_c, pc, n_parents, tid, pstride);
    ts.publish(tbl[0], ctb[0], tid);
    __syncthreads();
    issue(0);
    ts.fetch(patch, idx_c, pc + G, n_parents, tid, pstride);
    int buf = 0;
    for (; pc < n_parents; pc += G)
      {
        ts.publish(tbl[buf ^ 1], ctb[buf ^ 1], tid);
        T cur[COLOURED ? N : 1];
        if (COLOURED)
          {
#pragma unroll
            for (int i = 0; i < N; ++i)
              cur[COLOURED ? i : 0] = co[COLOURED ? i : 0];
          }
        // z^T on the prefetched registers: t2[k][b][a] = sum_c P1[c][k] w v[c]
#pragma unroll
        for (int it = 0; it < NL; ++it)
          {
            T r[N], xw[M];
#pragma unroll
            for (int c = 0; c < M; ++c)
              xw[c] = sh[it][zslot(c)] == 31u ? T(0) : v[it][c] * (T(1) / T(1u << sh[it][zslot(c)]));
            restrict_half<P, T>(pe, xw, r);
            if (has[it])
              {
                const int l = tid + it * TH;
#pragma unroll
                for (int k = 0; k < N; ++k)
                  t2[k * M * M + l] = r[k];
              }
          }
        __syncthreads();
        if (pc + G < n_parents)
          issue(buf ^ 1);
        ts.fetch(patch, idx_c, pc + 2 * G, n_parents, tid, pstride);
        // y^T: line (k, a)
        for (int l = tid; l < N * M; l += TH)
          {
            const int a = l % M, k = l / M;
            T         x[M], r[N];
#pragma unroll
            for (int b = 0; b < M; ++b)
              x[b] = t2[(k * M + b) * M + a];
            restrict_half<P, T>(pe, x, r);
#pragma unroll
            for (int j = 0; j < N; ++j)
              t1[(k * N + j) * M + a] = r[j];
          }
        __syncthreads();
        // x^T and scatter: line (j, k) of the coarse cell
        if (cl)
          {
            const int l = tid;
            T         x[M], r[N];
#pragma unroll
            for (int a = 0; a < M; ++a)
              x[a] = t1[l * M + a];
            restrict_half<P, T>(pe, x, r);
            const uint32_t b0 = ctb[buf][crow], b1 = ctb[buf][crow + 1], b2 = ctb[buf][crow + 2];
            if (COLOURED)
              {
                if (b0 != kInvalid)
                  coarse[b0 + coff] = cur[0] + r[0];
                if (b1 != kInvalid)
                  {
#pragma unroll
                    for (int i = 1; i < P; ++i)
                      coarse[b1 + coff * (uint32_t)(P - 1) + (uint32_t)(i - 1)] = cur[COLOURED ? i : 0] + r[i];
                  }
                if (b2 != kInvalid)
                  coarse[b2 + coff] = cur[COLOURED ? P : 0] + r[P];
              }
            else if (scratch)
              {
                T *o = scratch + (size_t)pc * (N * N * N) + (size_t)(l * N);
#pragma unroll
                for (int i = 0; i < N; ++i)
                  o[i] = r[i];
              }
            else
              {
                if (b0 != kInvalid)
                  unsafeAtomicAdd(&coarse[b0 + coff], r[0]);
                if (b1 != kInvalid)
                  {
#pragma unroll
                    for (int i = 1; i < P; ++i)
                      unsafeAtomicAdd(&coarse[b1 + coff * (uint32_t)(P - 1) + (uint32_t)(i - 1)], r[i]);
                  }
                if (b2 != kInvalid)
                  unsafeAtomicAdd(&coarse[b2 + coff], r[P]);
              }
          }
        __syncthreads(); // tables of `buf`, t1 and t2 are free again
        buf ^= 1;
      }
  }

  // ------------------------------------------------------------------------------------------
  // prolongate / prolongate_and_add: fine (+)= P coarse; every fine DoF is written by the one
  // parent that owns its entity (the values all parents compute for a shared DoF are bitwise
  // identical, see mgx_kernels.hip), so no atomics and bitwise reproducible.
  // ------------------------------------------------------------------------------------------
  template <int P, typename T, bool ADD>
  __global__ void __launch_bounds__(TPCfg<P>::THREADS)
    prolongate_pipe_kernel(T *__restrict__ fine, const T *__restrict__ coarse, const uint32_t *__restrict__ patch,
                           const uint32_t *__restrict__ idx_c, uint32_t n_parents, const Basis1D<T> *__restrict__ B)
  {
    using C          = TPCfg<P>;
    constexpr int N = C::N, M = C::M, TH = C::THREADS, NL = C::NL;
    constexpr int NA = ADD ? M : 1;
    static_assert(N * N <= TH, "one coarse x-line per thread");
    __shared__ uint32_t tbl[2][128];
    __shared__ uint32_t ctb[2][32];
    __shared__ T        t1[N * N * M]; // [k][j][a]
    __shared__ T        t2[N * M * M]; // [k][b][a]
    const int      tid = threadIdx.x;
    const uint32_t G   = gridDim.x;
    const T       *pe  = B->P1eo; // the embedding in even-odd form (mgx_brick_device.hpp prolong_half)

    PatchLine line[NL];
    bool      has[NL];
#pragma unroll
    for (int it = 0; it < NL; ++it)
      {
        const int l = tid + it * TH;
        has[it]     = l < M * M;
        line[it]    = patch_line<P>(has[it] ? l % M : 0, has[it] ? l / M : 0);
      }
    auto zslot = [](int c) {
      int e, o, n;
      patch_layer<P>(c, e, o, n);
      return e;
    };
    auto zoff = [](int c) {
      int e, o, n;
      patch_layer<P>(c, e, o, n);
      return o;
    };
    // coarse x-line (j, k) of this thread
    const bool cl  = tid < N * N;
    int        crow = 0;
    uint32_t   coff = 0;
    coarse_line<P>(cl ? tid % N : 0, cl ? tid / N : 0, crow, coff);

    uint32_t       pc = blockIdx.x;
    TableStage<TH> ts;
    T              cv[N], fo[NL][NA];
    auto           issue = [&](int b) {
      const uint32_t b0 = ctb[b][crow], b1 = ctb[b][crow + 1], b2 = ctb[b][crow + 2];
      const bool     v0 = cl && b0 != kInvalid, v1 = cl && b1 != kInvalid, v2 = cl && b2 != kInvalid;
      cv[0]             = coarse[v0 ? b0 + coff : 0u];
#pragma unroll
      for (int i = 1; i < P; ++i)
        cv[i] = coarse[v1 ? b1 + coff * (uint32_t)(P - 1) + (uint32_t)(i - 1) : 0u];
      cv[P] = coarse[v2 ? b2 + coff : 0u];
      if (!v0)
        cv[0] = T(0);
#pragma unroll
      for (int i = 1; i < P; ++i)
        if (!v1)
          cv[i] = T(0);
      if (!v2)
        cv[P] = T(0);
      if (ADD)
        {
#pragma unroll
          for (int it = 0; it < NL; ++it)
            {
              uint32_t w[5];
#pragma unroll
              for (int e = 0; e < 5; ++e)
                w[e] = tbl[b][has[it] ? 25 * e + line[it].sxy : 0];
#pragma unroll
              for (int c = 0; c < M; ++c)
                {
                  const uint32_t ww = w[zslot(c)];
                  fo[it][c % NA]    = fine[(has[it] && pw_owned(ww))
                                          ? pw_index(ww) + (uint32_t)(zoff(c) * line[it].nxy + line[it].oxy)
                                          : 0u];
                }
            }
        }
    };

    ts.fetch(patch, idx_c, pc, n_parents, tid);
    ts.publish(tbl[0], ctb[0], tid);
    __syncthreads();
    issue(0);
    ts.fetch(patch, idx_c, pc + G, n_parents, tid);
    int buf = 0;
    for (; pc < n_parents; pc += G)
      {
        ts.publish(tbl[buf ^ 1], ctb[buf ^ 1], tid);
        // x on the prefetched coarse line: t1[k][j][a] = sum_i P1[a][i] u[i]
        if (cl)
          {
            T f[M];
            prolong_half<P, T>(pe, cv, f);
#pragma unroll
            for (int a = 0; a < M; ++a)
              t1[tid * M + a] = f[a];
          }
        T fcur[NL][NA];
        if (ADD)
          {
#pragma unroll
            for (int it = 0; it < NL; ++it)
#pragma unroll
              for (int c = 0; c < NA; ++c)
                fcur[it][c] = fo[it][c];
          }
        __syncthreads();
        if (pc + G < n_parents)
          issue(buf ^ 1);
        ts.fetch(patch, idx_c, pc + 2 * G, n_parents, tid);
        // y: line (k, a)
        for (int l = tid; l < N * M; l += TH)
          {
            const int a = l % M, k = l / M;
            T         x[N];
#pragma unroll
            for (int j = 0; j < N; ++j)
              x[j] = t1[(k * N + j) * M + a];
            T f[M];
            prolong_half<P, T>(pe, x, f);
#pragma unroll
            for (int b = 0; b < M; ++b)
              t2[(k * M + b) * M + a] = f[b];
          }
        __syncthreads();
        // z into registers, owner writes
#pragma unroll
        for (int it = 0; it < NL; ++it)
          {
            if (!has[it])
              continue;
            const int l = tid + it * TH;
            T         x[N];
#pragma unroll
            for (int k = 0; k < N; ++k)
              x[k] = t2[k * M * M + l];
            uint32_t w[5];
#pragma unroll
            for (int e = 0; e < 5; ++e)
              w[e] = tbl[buf][25 * e + line[it].sxy];
            T f[M];
            prolong_half<P, T>(pe, x, f);
#pragma unroll
            for (int c = 0; c < M; ++c)
              {
                const uint32_t ww = w[zslot(c)];
                if (pw_owned(ww))
                  fine[pw_index(ww) + (uint32_t)(zoff(c) * line[it].nxy + line[it].oxy)] =
                    ADD ? fcur[it][c % NA] + f[c] : f[c];
              }
          }
        __syncthreads();
        buf ^= 1;
      }
  }


  // ------------------------------------------------------------------------------------------
  template <int P, typename T>
  static void launch_t(hipStream_t s, const TransferData &t, int what, void *fine, const void *coarse_in,
                       void *coarse_out, bool add, bool with_constraints)
  {
    using C               = TPCfg<P>;
    const OperatorData &c = *t.coarse;
    const uint32_t     *idx_c = with_constraints ? c.idx27 : c.idx27_plain;
    // persistent grid: as many workgroups as the device holds at once (queried per kernel)
    auto grid_of = [&](int slot, const void *kernel) {
      if (t.pipe_grid[slot] == 0)
        {
          // (one query at a time: eight ranks running as threads of one process asked for the same kernel at the same
          // moment and all got hipErrorUnknown from the runtime; a failed query must not leave its error behind for
          // the caller's hipGetLastError either)
          static std::mutex occupancy_mutex;
          int               per_cu = 1;
          {
            std::lock_guard<std::mutex> lock(occupancy_mutex);
            hipError_t                  e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, C::THREADS, 0);
            if (e != hipSuccess)
              {
                (void)hipGetLastError();
                e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, C::THREADS, 0);
              }
            if (e != hipSuccess || per_cu < 1)
              {
                (void)hipGetLastError();
                per_cu = 1;
              }
          }
          t.pipe_grid[slot] = (uint32_t)per_cu * t.n_cus;
        }
      return std::min<uint32_t>(c.n_cells, t.pipe_grid[slot]);
    };
    if (what == 0)
      {
        if (add)
          hipLaunchKernelGGL((prolongate_pipe_kernel<P, T, true>),
                             dim3(grid_of(0, (const void *)prolongate_pipe_kernel<P, T, true>)), dim3(C::THREADS), 0,
                             s, (T *)fine, (const T *)coarse_in, t.patch, idx_c, c.n_cells,
                             (const Basis1D<T> *)c.basis);
        else
          hipLaunchKernelGGL((prolongate_pipe_kernel<P, T, false>),
                             dim3(grid_of(1, (const void *)prolongate_pipe_kernel<P, T, false>)), dim3(C::THREADS),
                             0, s, (T *)fine, (const T *)coarse_in, t.patch, idx_c, c.n_cells,
                             (const Basis1D<T> *)c.basis);
      }
    else if (t.coarse_coloured && c.n_cells >= t.colour_min)
      {
        // 8 launches, one per colour (parent index mod 8): atomic-free and deterministic
        const uint32_t g = std::min<uint32_t>(c.n_cells / 8u, grid_of(3, (const void *)restrict_pipe_kernel<P, T, true>));
        for (uint32_t colour = 0; colour < 8; ++colour)
          hipLaunchKernelGGL((restrict_pipe_kernel<P, T, true>), dim3(g), dim3(C::THREADS), 0, s, (T *)coarse_out,
                             (const T *)fine, t.patch, idx_c, c.n_cells, (const Basis1D<T> *)c.basis, colour,
                             (T *)nullptr, t.owner_weights ? 1 : 0);
      }
    else
      {
        // coarse levels with the tables of the ordered assembly (built from the constrained index table):
        // per-parent sums to the scratch array, then coarse += their ordered sum; else atomic adds
        T *scratch = (c.asm_start && with_constraints) ? (T *)c.cell_scratch : nullptr;
        hipLaunchKernelGGL((restrict_pipe_kernel<P, T, false>),
                           dim3(grid_of(2, (const void *)restrict_pipe_kernel<P, T, false>)), dim3(C::THREADS), 0, s,
                           (T *)coarse_out, (const T *)fine, t.patch, idx_c, c.n_cells, (const Basis1D<T> *)c.basis,
                           0u, scratch, t.owner_weights ? 1 : 0);
        if (scratch)
          launch_assemble(s, c, 1, coarse_out, nullptr, 0u);
      }
  }

#define MGX_TP_DISPATCH(p, ...)                            \
  switch (p)                                               \
    {                                                      \
      case 1: { constexpr int P = 1; __VA_ARGS__; } break; \
      case 2: { constexpr int P = 2; __VA_ARGS__; } break; \
      case 3: { constexpr int P = 3; __VA_ARGS__; } break; \
      case 4: { constexpr int P = 4; __VA_ARGS__; } break; \
      case 5: { constexpr int P = 5; __VA_ARGS__; } break; \
      case 6: { constexpr int P = 6; __VA_ARGS__; } break; \
      case 7: { constexpr int P = 7; __VA_ARGS__; } break; \
      case 8: { constexpr int P = 8; __VA_ARGS__; } break; \
      case 9: { constexpr int P = 9; __VA_ARGS__; } break; \
      default: break;                                      \
    }

  void launch_prolongate_pipe(hipStream_t s, const TransferData &t, void *fine, const void *coarse, bool add,
                              bool with_constraints)
  {
    if (t.coarse->number == 1)
      {
        MGX_TP_DISPATCH(t.coarse->p, launch_t<P, double>(s, t, 0, fine, coarse, nullptr, add, with_constraints));
      }
    else
      {
        MGX_TP_DISPATCH(t.coarse->p, launch_t<P, float>(s, t, 0, fine, coarse, nullptr, add, with_constraints));
      }
  }

  void launch_restrict_add_pipe(hipStream_t s, const TransferData &t, void *coarse, const void *fine,
                                bool with_constraints)
  {
    if (t.coarse->number == 1)
      {
        MGX_TP_DISPATCH(t.coarse->p, launch_t<P, double>(s, t, 1, const_cast<void *>(fine), nullptr, coarse, false, with_constraints));
      }
    else
      {
        MGX_TP_DISPATCH(t.coarse->p, launch_t<P, float>(s, t, 1, const_cast<void *>(fine), nullptr, coarse, false, with_constraints));
      }
  }

} // namespace mgx
