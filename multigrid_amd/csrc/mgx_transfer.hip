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
#include "mgx_internal.hpp"

#include <hip/hip_runtime.h>

namespace mgx
{
  template <int P>
  struct TPCfg
  {
    static constexpr int N       = P + 1;
    static constexpr int M       = 2 * P + 1;
    static constexpr int M3      = M * M * M;
    static constexpr int N3      = N * N * N;
    static constexpr int THREADS = P <= 2 ? 64 : (P <= 4 ? 128 : 256);
    static constexpr int NIT     = (M3 + THREADS - 1) / THREADS; // patch points per thread
    static constexpr int NCT     = (N3 + THREADS - 1) / THREADS; // coarse values per thread
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
    if (a == 0)
      {
        layer = 0;
        off   = 0;
        size  = 1;
      }
    else if (a < P)
      {
        layer = 1;
        off   = a - 1;
        size  = P - 1;
      }
    else if (a == P)
      {
        layer = 2;
        off   = 0;
        size  = 1;
      }
    else if (a < 2 * P)
      {
        layer = 3;
        off   = a - P - 1;
        size  = P - 1;
      }
    else
      {
        layer = 4;
        off   = 0;
        size  = 1;
      }
  }

  // patch point q (lexicographic in the (2p+1)^3 patch) -> table slot | offset << 8; ~0 beyond
  template <int P>
  __device__ __forceinline__ uint32_t patch_code_of(int q)
  {
    constexpr int M = 2 * P + 1;
    if (q >= M * M * M)
      return 0xFFFFFFFFu;
    const int x = q % M, y = (q / M) % M, z = q / (M * M);
    int       ex, ey, ez, ox, oy, oz, nx, ny, nz;
    patch_layer<P>(x, ex, ox, nx);
    patch_layer<P>(y, ey, oy, ny);
    patch_layer<P>(z, ez, oz, nz);
    (void)nz;
    return (uint32_t)((ez * 5 + ey) * 5 + ex) | ((uint32_t)((oz * ny + oy) * nx + ox) << 8);
  }

  // coarse value c (lexicographic in the (p+1)^3 cell) -> slot in the 27-entry row | offset << 8
  template <int P>
  __device__ __forceinline__ uint32_t cell_code_of(int c)
  {
    constexpr int N = P + 1;
    if (c >= N * N * N)
      return 0xFFFFFFFFu;
    const int i = c % N, j = (c / N) % N, k = c / (N * N);
    const int cx = i == 0 ? 0 : (i == P ? 2 : 1), cy = j == 0 ? 0 : (j == P ? 2 : 1), cz = k == 0 ? 0 : (k == P ? 2 : 1);
    const int ox = cx == 1 ? i - 1 : 0, oy = cy == 1 ? j - 1 : 0, oz = cz == 1 ? k - 1 : 0;
    const int nx = cx == 1 ? P - 1 : 1, ny = cy == 1 ? P - 1 : 1;
    return (uint32_t)((cz * 3 + cy) * 3 + cx) | ((uint32_t)((oz * ny + oy) * nx + ox) << 8);
  }

  // ------------------------------------------------------------------------------------------
  // restrict_and_add: coarse += P^T (w .* fine), w = 1/multiplicity of the fine DoF among the
  // parent patches.  Shared coarse DoFs receive the parents' contributions by atomic adds.
  // ------------------------------------------------------------------------------------------
  template <int P, typename T>
  __global__ void __launch_bounds__(TPCfg<P>::THREADS)
    restrict_pipe_kernel(T *__restrict__ coarse, const T *__restrict__ fine, const uint32_t *__restrict__ patch,
                         const uint32_t *__restrict__ idx_c, uint32_t n_parents, const Basis1D<T> *__restrict__ B)
  {
    using C          = TPCfg<P>;
    constexpr int N = C::N, M = C::M, M3 = C::M3, TH = C::THREADS, NIT = C::NIT, NCT = C::NCT;
    __shared__ uint32_t tbl[2][128];
    __shared__ uint32_t ctb[2][32];
    __shared__ T        p1[M * N];
    __shared__ T        out[M3];
    __shared__ T        t2[N * M * M];
    __shared__ T        t1[N * N * M];
    const int      tid = threadIdx.x;
    const uint32_t G   = gridDim.x;
    for (int i = tid; i < M * N; i += TH)
      p1[i] = B->P1[i];
    uint32_t code[NIT], ccode[NCT];
#pragma unroll
    for (int it = 0; it < NIT; ++it)
      code[it] = patch_code_of<P>(tid + it * TH);
#pragma unroll
    for (int it = 0; it < NCT; ++it)
      ccode[it] = cell_code_of<P>(tid + it * TH);

    uint32_t pc = blockIdx.x;
    // table rows of a parent: threads 0..124 the patch row (two words per thread with 64
    // threads), threads 0..26 the coarse row
    uint32_t wreg = 0, wreg2 = 0, creg = 0;
    auto     fetch_tables = [&](uint32_t parent) {
      const bool ok = parent < n_parents;
      wreg          = (ok && tid < 125) ? patch[125u * (size_t)parent + tid] : 0u;
      if (TH < 125)
        wreg2 = (ok && tid + 64 < 125) ? patch[125u * (size_t)parent + tid + 64] : 0u;
      creg = (ok && tid < 27) ? idx_c[27u * (size_t)parent + tid] : kInvalid;
    };
    auto publish = [&](int b) {
      if (tid < 125)
        tbl[b][tid] = wreg;
      if (TH < 125 && tid + 64 < 125)
        tbl[b][tid + 64] = wreg2;
      if (tid < 27)
        ctb[b][tid] = creg;
    };
    T        v[NIT];
    uint64_t shifts = 0;
    auto     issue  = [&](int b) {
      shifts = 0;
#pragma unroll
      for (int it = 0; it < NIT; ++it)
        {
          const uint32_t cd = code[it];
          const uint32_t w  = tbl[b][cd == 0xFFFFFFFFu ? 0 : (cd & 0xFF)];
          const uint32_t a  = cd == 0xFFFFFFFFu ? 0u : pw_index(w) + (cd >> 8);
          v[it]             = fine[a];
          shifts |= (uint64_t)pw_shift(w) << (2 * it);
        }
    };

    fetch_tables(pc);
    publish(0);
    __syncthreads();
    issue(0);
    fetch_tables(pc + G);
    int buf = 0;
    for (; pc < n_parents; pc += G)
      {
        publish(buf ^ 1);
        // weighted fine values of this parent (loaded during the previous iteration)
#pragma unroll
        for (int it = 0; it < NIT; ++it)
          if (code[it] != 0xFFFFFFFFu)
            {
              const uint32_t sh = (uint32_t)(shifts >> (2 * it)) & 3u;
              out[tid + it * TH] = v[it] * (T(1) / T(1u << sh));
            }
        __syncthreads();
        if (pc + G < n_parents)
          issue(buf ^ 1);
        fetch_tables(pc + 2 * G);
        for (int o = tid; o < N * M * M; o += TH) // z^T: [k][b][a]
          {
            const int ba = o % (M * M), k = o / (M * M);
            T         s  = 0;
#pragma unroll
            for (int c = 0; c < M; ++c)
              s = fma(p1[c * N + k], out[c * M * M + ba], s);
            t2[o] = s;
          }
        __syncthreads();
        for (int o = tid; o < N * N * M; o += TH) // y^T: [k][j][a]
          {
            const int a = o % M, j = (o / M) % N, k = o / (M * N);
            T         s = 0;
#pragma unroll
            for (int b = 0; b < M; ++b)
              s = fma(p1[b * N + j], t2[(k * M + b) * M + a], s);
            t1[o] = s;
          }
        __syncthreads();
        // x^T and scatter: thread c owns coarse value (i, j, k)
#pragma unroll
        for (int it = 0; it < NCT; ++it)
          {
            const uint32_t cd = ccode[it];
            if (cd == 0xFFFFFFFFu)
              continue;
            const int c = tid + it * TH, i = c % N, kj = c / N;
            T         s = 0;
#pragma unroll
            for (int a = 0; a < M; ++a)
              s = fma(p1[a * N + i], t1[kj * M + a], s);
            const uint32_t w = ctb[buf][cd & 0xFF];
            if (w != kInvalid)
              unsafeAtomicAdd(&coarse[w + (cd >> 8)], s);
          }
        __syncthreads(); // tables of `buf` and the sweep buffers are free again
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
    constexpr int N = C::N, M = C::M, N3 = C::N3, TH = C::THREADS, NIT = C::NIT, NCT = C::NCT;
    __shared__ uint32_t tbl[2][128];
    __shared__ uint32_t ctb[2][32];
    __shared__ T        p1[M * N];
    __shared__ T        in[N3];
    __shared__ T        t1[N * N * M];
    __shared__ T        t2[N * M * M];
    const int      tid = threadIdx.x;
    const uint32_t G   = gridDim.x;
    for (int i = tid; i < M * N; i += TH)
      p1[i] = B->P1[i];
    uint32_t code[NIT], ccode[NCT];
#pragma unroll
    for (int it = 0; it < NIT; ++it)
      code[it] = patch_code_of<P>(tid + it * TH);
#pragma unroll
    for (int it = 0; it < NCT; ++it)
      ccode[it] = cell_code_of<P>(tid + it * TH);

    uint32_t pc = blockIdx.x;
    uint32_t wreg = 0, wreg2 = 0, creg = 0;
    auto     fetch_tables = [&](uint32_t parent) {
      const bool ok = parent < n_parents;
      wreg          = (ok && tid < 125) ? patch[125u * (size_t)parent + tid] : 0u;
      if (TH < 125)
        wreg2 = (ok && tid + 64 < 125) ? patch[125u * (size_t)parent + tid + 64] : 0u;
      creg = (ok && tid < 27) ? idx_c[27u * (size_t)parent + tid] : kInvalid;
    };
    auto publish = [&](int b) {
      if (tid < 125)
        tbl[b][tid] = wreg;
      if (TH < 125 && tid + 64 < 125)
        tbl[b][tid + 64] = wreg2;
      if (tid < 27)
        ctb[b][tid] = creg;
    };
    T    cv[NCT], fo[ADD ? NIT : 1];
    auto issue = [&](int b) {
#pragma unroll
      for (int it = 0; it < NCT; ++it)
        {
          const uint32_t cd    = ccode[it];
          const uint32_t w     = ctb[b][cd == 0xFFFFFFFFu ? 0 : (cd & 0xFF)];
          const bool     valid = cd != 0xFFFFFFFFu && w != kInvalid;
          const T        x     = coarse[valid ? w + (cd >> 8) : 0u];
          cv[it]               = valid ? x : T(0);
        }
      if (ADD)
        {
#pragma unroll
          for (int it = 0; it < NIT; ++it)
            {
              const uint32_t cd = code[it];
              const uint32_t w  = tbl[b][cd == 0xFFFFFFFFu ? 0 : (cd & 0xFF)];
              const bool     mine = cd != 0xFFFFFFFFu && pw_owned(w);
              fo[ADD ? it : 0]  = fine[mine ? pw_index(w) + (cd >> 8) : 0u];
            }
        }
    };

    fetch_tables(pc);
    publish(0);
    __syncthreads();
    issue(0);
    fetch_tables(pc + G);
    int buf = 0;
    for (; pc < n_parents; pc += G)
      {
        publish(buf ^ 1);
#pragma unroll
        for (int it = 0; it < NCT; ++it)
          if (ccode[it] != 0xFFFFFFFFu)
            in[tid + it * TH] = cv[it];
        // the old fine values of this parent stay in registers until the z sweep; move them out of
        // the way of the next parent's prefetch
        T fcur[ADD ? NIT : 1];
        if (ADD)
          {
#pragma unroll
            for (int it = 0; it < NIT; ++it)
              fcur[ADD ? it : 0] = fo[ADD ? it : 0];
          }
        __syncthreads();
        if (pc + G < n_parents)
          issue(buf ^ 1);
        fetch_tables(pc + 2 * G);
        for (int o = tid; o < N * N * M; o += TH) // x: [k][j][a]
          {
            const int a = o % M, kj = o / M;
            T         s = 0;
#pragma unroll
            for (int i = 0; i < N; ++i)
              s = fma(p1[a * N + i], in[kj * N + i], s);
            t1[o] = s;
          }
        __syncthreads();
        for (int o = tid; o < N * M * M; o += TH) // y: [k][b][a]
          {
            const int a = o % M, b = (o / M) % M, k = o / (M * M);
            T         s = 0;
#pragma unroll
            for (int j = 0; j < N; ++j)
              s = fma(p1[b * N + j], t1[(k * N + j) * M + a], s);
            t2[o] = s;
          }
        __syncthreads();
        // z sweep into registers, owner writes
#pragma unroll
        for (int it = 0; it < NIT; ++it)
          {
            const uint32_t cd = code[it];
            if (cd == 0xFFFFFFFFu)
              continue;
            const int o = tid + it * TH, ba = o % (M * M), c = o / (M * M);
            T         s = 0;
#pragma unroll
            for (int k = 0; k < N; ++k)
              s = fma(p1[c * N + k], t2[k * M * M + ba], s);
            const uint32_t w = tbl[buf][cd & 0xFF];
            if (pw_owned(w))
              fine[pw_index(w) + (cd >> 8)] = ADD ? fcur[ADD ? it : 0] + s : s;
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
    const uint32_t      grid  = std::min<uint32_t>(c.n_cells, t.pipe_grid);
    if (what == 0)
      {
        if (add)
          hipLaunchKernelGGL((prolongate_pipe_kernel<P, T, true>), dim3(grid), dim3(C::THREADS), 0, s, (T *)fine,
                             (const T *)coarse_in, t.patch, idx_c, c.n_cells, (const Basis1D<T> *)c.basis);
        else
          hipLaunchKernelGGL((prolongate_pipe_kernel<P, T, false>), dim3(grid), dim3(C::THREADS), 0, s, (T *)fine,
                             (const T *)coarse_in, t.patch, idx_c, c.n_cells, (const Basis1D<T> *)c.basis);
      }
    else
      hipLaunchKernelGGL((restrict_pipe_kernel<P, T>), dim3(grid), dim3(C::THREADS), 0, s, (T *)coarse_out,
                         (const T *)fine, t.patch, idx_c, c.n_cells, (const Basis1D<T> *)c.basis);
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

  // workgroups per CU the kernels can hold (LDS bound), for the persistent grid
  int transfer_pipe_blocks_per_cu(int p, int number)
  {
    const size_t ts = number == 1 ? 8 : 4;
    const size_t N = p + 1, M = 2 * p + 1;
    const size_t lds = 4 * (2 * 128 + 2 * 32) + ts * (M * N + M * M * M + N * M * M + N * N * M);
    const int    th  = p <= 2 ? 64 : (p <= 4 ? 128 : 256);
    const int    by_lds = (int)std::max<size_t>(1, (size_t)(160 * 1024) / lds);
    const int    by_waves = 32 / (th / 64);
    return std::min(by_lds, by_waves);
  }
} // namespace mgx
