// mgx_internal.hpp -- internal C++ interface between the C ABI (mgx_api.cpp) and the HIP kernels
// (mgx_kernels.hip).  Not installed; the public boundary is include/mgx.h.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

struct mgx_context_s;

namespace mgx
{
  constexpr int      kMaxN    = 10; // p <= 9
  constexpr uint32_t kInvalid = 0xFFFFFFFFu;

  // Options of a context, carried by the objects built on it (two contexts of one process can differ:
  // the tests compare code paths that way).  The thresholds marked ENV are also read from the
  // environment, ONCE, when a context is created; everything else is set only through
  // mgx_context_set_option(ctx, "<name>", value) before the first object is created on the context.
  // All but rccl_selftest select between numerically equivalent code paths or set thresholds.
  struct Tunables
  {
    bool     trace            = false; // MGX_TRACE            host control flow on stderr
    bool     general_kernel   = false; // MGX_GENERAL_KERNEL   quadrature-point form instead of the separable one
    bool     no_bricks        = false; // MGX_NO_BRICKS        per-cell kernel on every level
    uint32_t brick_min        = 512;   // MGX_BRICK_MIN        bricks per level from which the brick loop is used (measured crossover with the per-cell kernel: between 216 and 512 bricks)
    bool     brick_min_from_env = false;
    uint32_t overlap_min      = 16384; // MGX_OVERLAP_MIN_BRICKS  bricks per rank from which interface bricks run first
    bool     cells_form       = false; // MGX_BRICK_FORM=cells cell-by-cell brick kernel instead of the macro-element one
    uint32_t wide_max         = 1024;  // MGX_BRICK_WIDE_MAX   cell-by-cell form: launches below this use 512 threads
    bool     dg_no_overlap    = false; // MGX_DG_NO_OVERLAP        DG ghost exchange before all cells instead of under the interior ones
    bool     no_restrict_scratch = false; // fused residual + restriction adds into the coarse vector colour by colour instead of through per-brick blocks
    bool     no_fused_residual = false;  // V-cycle: residual and restriction as separate kernels, the prolongation form stays fused
    bool     no_fused_assembly = false;  // per-cell levels: Chebyshev update as a kernel after the assembly kernel
    bool     no_fused_decomposed = false; // decomposed levels: residual / restriction / prolongation as separate kernels
    bool     dg_unmerged_restrict = false; // DG V-cycle: residual and DG -> FE_Q restriction as two kernels instead of the merged action 1
    uint32_t macro_wg_x16     = 0;     // MGX_MACRO_WG_PER_CU_X16  macro kernel grid, in 1/16 workgroups per CU [resident]
    bool     no_diag_table    = false; // MGX_NO_DIAG_TABLE    stream the inverse diagonal in the fused Chebyshev forms
    bool     roctx            = false; // profiler ranges with the reference's LIKWID region names (mgx_range_push/pop, per-level phases of the V-cycle)
    uint32_t fused_prolong_min_bricks = 8192; // p <= 4, levels on a reduced-colour schedule with fewer bricks (per rank): prolongation as a kernel of its own + first post-smoothing step on that schedule (17 M-DoF level of C2: 0.894 against 0.905 ms; emulated rank at 8 GPUs 2.417 against 2.446 ms; a 67 M-DoF rank level, 16 384 bricks, is faster fused: 5.42 against 5.48 ms; 0: fused everywhere)
    bool     no_general_bricks = false; // general tensor branch: per-cell kernel + ordered assembly also where the brick form exists (A/B, tests)
    uint32_t general_brick_min = 2048;  // ... bricks from which vmult of a general operator runs in brick form (one workgroup per brick, 512 resident: below four rounds of workgroups the per-cell kernel with its 16 waves per CU is faster)
    bool     no_macro_v2      = false; // first pipeline of the macro-element kernel (gather after the sweeps) for every form; A/B of mgx_macro2.hip
    bool     no_fused_init    = false; // MGX_NO_FUSED_INIT    store the first Chebyshev iterate
    bool     no_fused_restrict = false; // MGX_NO_FUSED_RESTRICT  separate residual and restriction kernels
    bool     no_fused_prolong = false; // MGX_NO_FUSED_PROLONG prolongation as a kernel of its own
    bool     force_fused_transfers = false; // (no-op since round 4: the fused transfer forms run at every degree)
    bool     transfer_v1      = false; // MGX_TRANSFER_V1      first-version transfer kernels
    bool     restrict_atomic  = false; // MGX_RESTRICT_ATOMIC  one-launch restriction with atomics on every level
    uint32_t restrict_colour_min = 16384; // MGX_RESTRICT_COLOUR_MIN  coarse cells from which restriction runs by colour
    bool     exchange_unfused = false; // MGX_EXCHANGE_UNFUSED one pack / unpack launch per neighbour
    uint32_t cell_colour_min  = 0xFFFFFFFFu; // MGX_CELL_COLOUR_MIN  general-coefficient levels from this many cells run colour by colour; by default only levels whose ordered-assembly tables would not fit do (measured on the shell meshes: one launch + ordered assembly 20 / 56 / 119 / 463 us at 6 k / 25 k / 49 k / 197 k cells against 103 / 136 / 202 / 540 us for the 13 colour launches of the greedy colouring)
    uint32_t free_max_bricks  = 16384; // MGX_FREE_MAX_BRICKS  levels with at most this many bricks run the plain / residual / Chebyshev forms of the brick loop on a reduced-colour schedule (mgx_macro.hip, FREE) instead of eight colour launches
    uint32_t free_one_max     = 1024;  // MGX_FREE_ONE_MAX     ... with one class (one launch) up to this many bricks, two classes beyond
    bool     no_graph         = false; // MGX_NO_GRAPH         no HIP-graph replay of the coarse levels
    uint32_t graph_max_dofs   = 600000; // MGX_GRAPH_MAX_DOFS  largest level inside the replayed graph
    bool     rccl_selftest    = false; // MGX_RCCL_SELFTEST    one-rank communicator may name itself as neighbour
    static Tunables from_environment();
    bool            set(const std::string &name, double value); // false: unknown option
  };

  // 1D data of the element in the operator's number type, resident in device memory and read
  // through wave-uniform (scalar) loads.
  // Even-odd (Appendix B of SURVEY.md, matrix_vector_kernel.h:47-113) form of a symmetric and
  // persymmetric n x n matrix A (A[a][b] = A[b][a] = A[n-1-a][n-1-b]), H = n/2:
  //   ee[a*H+i] = (A[a][i] + A[a][n-1-i])/2, eo[a*H+i] = (A[a][i] - A[a][n-1-i])/2  (a,i < H)
  //   mc[a] = A[a][H] (odd n: middle column = middle row), mhh = A[H][H]
  template <typename T>
  struct EOMat
  {
    T ee[(kMaxN / 2) * (kMaxN / 2)];
    T eo[(kMaxN / 2) * (kMaxN / 2)];
    T mc[kMaxN / 2];
    T mhh;
  };

  template <typename T>
  struct Basis1D
  {
    T S[kMaxN * kMaxN];       // S[q*n+i]   nodal -> quadrature
    T D[kMaxN * kMaxN];       // D[q*n+r]   collocation derivative
    T w[kMaxN];               // quadrature weights
    T P1[2 * kMaxN * kMaxN];  // P1[a*n+i]  prolongation, a in [0,2p]
    // 1D mass and stiffness matrices of the separable (Cartesian, constant coefficient) cell
    // matrix  A_cell = sum_d c_d (M x M x K_d):  M = S^T W S,  K = S^T D^T W D S
    EOMat<T> mass, lapl;
    // P1 in even-odd form for the line products of the fused transfer forms (restrict_half / prolong_line,
    // mgx_brick_device.hpp).  The embedding of a parent into its two children is symmetric under reversal of both
    // indices, P1[2p-a][p-j] = P1[a][j] (checked when the transfer is created).  With nh = (p+1)/2 pairs (j, p-j):
    //   HE[a*nh+j] = (P1[a][j] + P1[a][p-j]) / 2, a <= p;  HO[a*nh+j] = (P1[a][j] - P1[a][p-j]) / 2, a < p;
    //   PC[a] = P1[a][p/2], a <= p (p even: the coarse point in the middle of the parent)
    // at offsets 0, (p+1) nh, (2p+1) nh.
    T P1eo[2 * kMaxN * kMaxN];
  };

  // Reduced-colour schedule of the macro-element kernel (mgx_macro.hip, FREE; built by
  // build_free_schedule, mgx_bricks.cpp): one or two classes of bricks instead of eight colours.  The
  // entities bricks of one class share are PRIVATE: every brick stores their partial sums in a block
  // of its own, priv[brick][n_surf], and launch_surf_finish adds the blocks up per DoF (entries
  // surf_pos[surf_start[i] .. surf_start[i+1]) for DoF surf_dof[i], ascending) and applies the
  // post-operation.  The first n_surf_shared DoFs of that list are shared with other ranks: their
  // sum goes to the carrier vector and is completed after the exchange.
  struct FreeSchedule
  {
    int       n_classes = 0;                    // 1 or 2
    int       n_groups = 0, n_iface_groups = 0; // launch groups (classes, interface bricks first if split)
    uint32_t  group_start[9] = {0};
    uint32_t *ent         = nullptr; // device [n_bricks * entities]: entity table in group order, flags of this schedule
    uint32_t *surf_off    = nullptr; // device [entities per brick]: offset of a private entity in a block, kInvalid: not private
    uint32_t  n_surf      = 0;       // values per block
    void     *priv        = nullptr; // device, number type
    uint32_t *surf_dof    = nullptr, *surf_start = nullptr, *surf_pos = nullptr;
    uint32_t  n_surf_dofs = 0, n_surf_shared = 0;
    bool      available() const { return ent != nullptr; }
  };

  // Brick schedule of a level (mgx_brick.hip): 64-cell bricks sorted by colour, one launch per
  // colour; per brick the first DoF of each of its 9^3 mesh entities and the FIRST/LAST flags.
  struct BrickData
  {
    uint32_t  n_bricks  = 0;
    int       n_colours = 0;       // launch groups
    int       n_iface_groups = 0;  // leading groups = bricks on the rank interface (0: not split)
    uint32_t  colour_start[33] = {0};
    uint32_t *ent_base  = nullptr; // device [n_bricks * 729]
    uint8_t  *ent_flags = nullptr; // device [n_bricks * 729]  bit0 FIRST, bit1 LAST
    uint32_t *item_map  = nullptr; // device [(NB p + 1)^3]: write-out order of the macro-element kernel
    uint32_t *item_map2 = nullptr; // device: the same items, interior of the brick first (second pipeline, mgx_macro2.hip)
    std::vector<uint32_t> order; // host: colour-sorted position -> brick index in cell order
    uint32_t *order_dev = nullptr; // device copy (general operator: brick_general_kernel finds its cells through it)
    bool      available() const { return n_bricks > 0; }
    FreeSchedule fr; // reduced-colour schedule of the plain / residual / Chebyshev forms (may be absent)
  };
  constexpr int kBrickEntities = 729;
  constexpr int kMaxColours    = 32;

  // Device-side view of a LaplaceOperator level (laplace_operator.h:126-163)
  struct OperatorData
  {
    int       p        = 0;
    int       number   = 1; // MGX_F64
    uint32_t  n_cells  = 0;
    uint32_t  n_dofs   = 0;
    uint32_t  n_constrained = 0;
    uint32_t *idx27         = nullptr; // device
    uint32_t *idx27_plain   = nullptr; // device (may be null)
    uint32_t *constrained   = nullptr; // device
    void     *basis         = nullptr; // device Basis1D<T>
    void     *inv_diag      = nullptr; // device, number type
    double    coef[6]       = {0, 0, 0, 0, 0, 0};
    bool      full_tensor   = false;   // affine cells with off-diagonal coefficient entries (:473-486)
    void     *coef_q        = nullptr; // device [n_cells][6][n^3], number type: general branch (:493-522)
    void     *grad_1d       = nullptr; // device [n*n], number type: G = D S, nodal derivative at the quadrature points
    // general branch on levels with many cells: cells sorted by colour (cells of one colour share
    // no DoF), one launch per colour without atomics; nullptr: one launch with atomic adds
    uint32_t *cell_order    = nullptr; // device [n_cells]
    int       n_cell_colours = 0;
    uint32_t  cell_colour_start[33] = {0};
    // Ordered assembly of the per-cell kernels (levels without a brick schedule and without cell
    // colours): the kernels store each cell's (p+1)^3 local results in cell_scratch and
    // assemble_kernel adds, for every DoF d, the entries asm_pos[asm_start[d] .. asm_start[d+1]) in
    // ascending cell order -- no atomics, bitwise reproducible.  nullptr: not built.
    uint32_t *asm_start     = nullptr; // device [n_dofs + 1]
    uint32_t *asm_pos       = nullptr; // device: cell (p+1)^3 + local index (k n + j) n + i
    void     *cell_scratch  = nullptr; // device [n_cells (p+1)^3], number type
    BrickData bricks;
    BrickData gbricks; // general tensor branch, p = 4, one rank: one-launch schedule of brick_general_kernel (fr, item_map)
    bool      cells_form    = false; // Tunables::cells_form of the context the operator was created on
    uint32_t  wide_max      = 1024;  // Tunables::wide_max
    uint32_t  macro_wg_x16  = 0;     // Tunables::macro_wg_x16
    bool      separable     = true; // Cartesian constant-coefficient fast path of the brick loop
    // inverse diagonal per brick item ((NB p + 1)^3 values in item order) if it is the same for
    // every brick (uniform mesh), else nullptr: the macro-element kernel then reads it from
    // registers instead of streaming inv_diag (mgx_macro.hip, DTAB)
    void     *diag_items    = nullptr;
    void     *diag_items2   = nullptr; // ... in the order of bricks.item_map2 (second pipeline)
    bool      macro_v2      = true;    // !Tunables::no_macro_v2
  };

  struct TransferData
  {
    const OperatorData *coarse = nullptr, *fine = nullptr;
    uint32_t           *children = nullptr;    // device [n_coarse_cells*8]
    uint8_t            *weight_shift = nullptr; // device [n_coarse_cells*27]: weight = 2^-shift
    uint32_t           *own27 = nullptr;        // device [n_fine_cells]: bit e set iff the fine cell is
                                                // the first (in cell order) containing its entity e
    // pipelined transfers (mgx_transfer.hip): 125 words per parent for the 5^3 mesh entities of its
    // children patch: first fine DoF | log2(multiplicity) << 29 | owned-by-this-parent << 31;
    // nullptr if the fine level has 2^29 DoFs or more (first-version kernels are used then)
    uint32_t           *patch = nullptr;
    // fused residual + restriction (mgx_brick.hip, mode 7): per fine brick, in the fine level's
    // colour-sorted brick order, the first coarse DoF (constrained: invalid) of the (2 PB + 1)^3 mesh
    // entities of the PB^3 parents the brick's cells belong to (PB = 2 for p <= 4, 1 for p >= 5)
    uint32_t           *coarse_blocks = nullptr;
    // ... restricted values per brick, [n_bricks][(PB p + 1)^3] in the number type, and the ordered assembly of the
    // coarse vector from them: coarse[d] = sum of coarse_scratch[cs_pos[k]], k in [cs_start[d], cs_start[d + 1]),
    // in ascending brick order (launch_coarse_assemble).  The bricks of a level then write disjoint addresses and the
    // fused residual + restriction is ONE launch per level.
    void               *coarse_scratch = nullptr;
    uint32_t           *cs_start = nullptr, *cs_pos = nullptr;
    // ... on a decomposed mesh: the DoFs on the rank interface, which no brick completes, are transferred by two
    // list kernels after the exchange.  Restriction (CSR by coarse DoF over the interface DoFs this rank owns):
    // coarse[ifr_cdof[i]] += sum_k ifr_w[k] r[ifr_fdof[k]], k in [ifr_start[i], ifr_start[i+1]).  Prolongation (CSR by
    // entry i of the fine plan's list of shared DoFs): (P e)[shared[i]] = sum_k ifp_w[k] e[ifp_cdof[k]]
    uint32_t           *ifr_cdof = nullptr, *ifr_start = nullptr, *ifr_fdof = nullptr;
    void               *ifr_w = nullptr;
    uint32_t            n_ifr = 0;
    uint32_t           *ifp_start = nullptr, *ifp_cdof = nullptr;
    void               *ifp_w = nullptr;
    uint32_t            n_ifp = 0;
    mutable uint32_t    pipe_grid[4] = {0, 0, 0, 0}; // persistent grid of prolongate(add), prolongate, restrict x2
    bool                owner_weights = false;   // restriction: weight 1 for the parent that owns a fine entity, 0 for the others (multi-block meshes)
    bool                coarse_coloured = false; // coarse cells c and c' with c % 8 == c' % 8 share no DoF
    uint32_t            n_cus = 256;
    uint32_t            colour_min = 16384; // Tunables::restrict_colour_min
  };

  // transport of the context (mgx_api.cpp) for other translation units: exchange of packed device
  // buffers (counts[k] entries of `number` type with rank ranks[k], both directions), sum of a few
  // host doubles over the ranks (no-op on a single rank)
  int  exchange_buffers(struct ::mgx_context_s *ctx, int plan_id, int number, int n_neighbors, const int *ranks,
                        const uint32_t *counts, void *const *send, void *const *recv, hipStream_t stream = nullptr);
  // Overlap of an exchange with work on the context's stream: side_stream_begin returns the context's side stream, ordered
  // behind everything enqueued on the main stream so far (nullptr: no side stream, run in order);
  // side_stream_end orders the main stream behind the side stream again.
  hipStream_t side_stream_begin(struct ::mgx_context_s *ctx);
  int         side_stream_end(struct ::mgx_context_s *ctx);
  int  allreduce_sum(struct ::mgx_context_s *ctx, double *values, int count);
  int  dot_owned_prefix(struct ::mgx_context_s *ctx, int number, const void *x, const void *y, size_t n, double *out);
  bool context_has_comm(struct ::mgx_context_s *ctx);
  const Tunables &context_tunables(struct ::mgx_context_s *ctx);

  // records the message mgx_last_error() returns on the calling thread; returns `code` (used by the
  // translation units that implement parts of the C ABI outside mgx_api.cpp)
  int report_error(int code, const char *message);

  void launch_prolongate_pipe(hipStream_t s, const TransferData &t, void *fine, const void *coarse, bool add,
                              bool with_constraints);
  void launch_restrict_add_pipe(hipStream_t s, const TransferData &t, void *coarse, const void *fine,
                                bool with_constraints);

  // ---- cell loops (mgx_kernels.hip) ----
  // dst += A_cells * src  (MatrixFree::cell_loop(local_apply), laplace_operator.h:527-558)
  // op.asm_start (ordered assembly): dst is WRITTEN (no zeroing by the caller), and dst[d] = tail_src[d] for
  // d >= n_head if tail_src is given; otherwise dst must be zero on entry
  // post (ordered assembly only): the Chebyshev update of PreconditionChebyshev inside the assembly kernel instead of
  // a kernel of its own -- src is the iterate x, updated in place after every cell has read it:
  //   x_new = x + f2 dinv (b - A x) [+ f1 (x - x_old), three_term];  x_old = x;  rows >= n_head: A x = x (identity rows)
  struct ChebPost
  {
    void       *x_old;
    const void *b, *dinv;
    double      f1, f2;
    bool        three_term;
  };
  void launch_cell_loop(hipStream_t s, const OperatorData &op, void *dst, const void *src, const void *tail_src = nullptr,
                        uint32_t n_head = 0,
                        const ChebPost *post = nullptr);
  // general operator, brick form (p = 4, OperatorData::gbricks; mgx_kernels.hip brick_general_kernel)
  void launch_general_bricks(hipStream_t s, const OperatorData &op, void *dst, const void *src);
  // mode 0: dst = ordered sums of op.cell_scratch (tail as above), mode 1: dst += them
  void launch_assemble(hipStream_t s, const OperatorData &op, int mode, void *dst, const void *tail_src, uint32_t n_head);
  // brick cell loop with fused post-operation (mgx_brick.hip); mode = BrickMode
  //   0: out = A src          1: out = a - A src
  //   2: out = src + f1 (src - out) + f2 b (a - A src)      3: same without the f1 term
  // `partial` carries partial sums of brick-surface DoFs between the colour launches
  //   old: previous iterate of mode 2 (nullptr: it is `out`, which is then read before written)
  // group_begin / group_end: launch groups to run (default: all; the interface / interior halves of
  // a split schedule are launched separately, BrickData::n_iface_groups)
  void launch_brick_loop(hipStream_t s, const OperatorData &op, int mode, const void *src, const void *a,
                         const void *b, void *out, void *partial, double f1, double f2, const void *old = nullptr,
                         double f0 = 0., void *coarse = nullptr, const uint32_t *coarse_blocks = nullptr,
                         int group_begin = 0, int group_end = -1, bool free_schedule = false);
  // LaplaceOperator::compute_residual (laplace_operator.h:804-845) through the general per-cell kernel:
  // dst += sum over the cells of  S^T [ rhs_q - D^T (K D S (-src)) ], src read through idx27_plain; assembly as in
  // launch_cell_diagonal (lists of cells that share no DoF / ordered assembly / atomics).  dst zeroed by the caller.
  void launch_cell_residual(hipStream_t s, const OperatorData &op, void *dst, const void *src, const void *rhs_q,
                            const uint32_t *lists = nullptr, const uint32_t *list_start = nullptr, int n_lists = 0);
  // free_schedule (modes 0..6, op.bricks.fr.available()): the launch groups are those of the reduced-colour
  // schedule; the caller completes the private DoFs with launch_surf_finish: DoFs [first, first + count)
  // of op.bricks.fr.surf_dof; those below n_surf_shared: carrier[d] = sum only
  // constrained / n_constrained (Chebyshev forms): rows where A x = x, updated by the same launch
  void launch_surf_finish(hipStream_t s, const OperatorData &op, int mode, uint32_t first, uint32_t count, void *carrier,
                          const void *x, void *out, const void *a, const void *dinv, const void *old, double f1, double f2,
                          double f0, const uint32_t *constrained = nullptr, uint32_t n_constrained = 0,
                          const FreeSchedule *schedule = nullptr); // schedule: another one than op.bricks.fr
  // macro-element form of the separable brick loop (mgx_macro.hip), one translation unit per number
  // type; false: mode / degree not covered (the caller falls back to the cell-by-cell form)
  bool launch_macro_loop_f64(hipStream_t s, const OperatorData &op, int mode, const void *src, const void *a,
                             const void *b, void *out, void *partial, double f1, double f2, const void *old,
                             double f0, void *coarse, const uint32_t *coarse_blocks, int group_begin, int group_end,
                             bool free_schedule);
  bool launch_macro_loop_f32(hipStream_t s, const OperatorData &op, int mode, const void *src, const void *a,
                             const void *b, void *out, void *partial, double f1, double f2, const void *old,
                             double f0, void *coarse, const uint32_t *coarse_blocks, int group_begin, int group_end,
                             bool free_schedule);
  // fused PCG step on a brick-scheduled level (mgx_macro.hip, kCgUpdate)
  bool launch_macro_cg_update_f64(hipStream_t s, const OperatorData &op, double alpha, double beta, const void *r, void *q,
                                  void *p, void *x, void *carrier, double *partials, uint32_t capacity,
                                  uint32_t *n_partials);
  bool launch_macro_cg_update_f32(hipStream_t s, const OperatorData &op, double alpha, double beta, const void *r, void *q,
                                  void *p, void *x, void *carrier, double *partials, uint32_t capacity,
                                  uint32_t *n_partials);
  void launch_reduce4(hipStream_t s, const double *partials, uint32_t n, const double *extra, double *sums);
  void macro_diag_table_f64(hipStream_t s, const OperatorData &op, const uint32_t *item_map, void *table, uint32_t *flag_dev);
  void macro_diag_table_f32(hipStream_t s, const OperatorData &op, const uint32_t *item_map, void *table, uint32_t *flag_dev);
  // second pipeline (mgx_macro2.hip: plain, residual, residual + restriction); false: form / degree not covered, use the first
  bool launch_macro2_loop_f64(hipStream_t s, const OperatorData &op, int mode, const void *src, const void *a, void *out,
                              void *partial, void *coarse, const uint32_t *coarse_blocks, int group_begin, int group_end,
                              double f1, double f2, double f0, const void *old);
  bool launch_macro2_loop_f32(hipStream_t s, const OperatorData &op, int mode, const void *src, const void *a, void *out,
                              void *partial, void *coarse, const uint32_t *coarse_blocks, int group_begin, int group_end,
                              double f1, double f2, double f0, const void *old);
  // true: the brick loop evaluates the separable form (7 sweeps); false: the general
  // quadrature-point form of laplace_operator.h:436-523 (12 sweeps)
  // diag += diagonal of the cell matrices (local_compute_diagonal, laplace_operator.h:770-800)
  // a1d[i] = sum_q w_q (dphi_i(x_q))^2, m1d[i] = sum_q w_q phi_i(x_q)^2 (host arrays, n entries)
  // lists / list_start / n_lists: device cell lists of launches whose cells share no DoF (plain adds);
  // n_lists == 0: ordered assembly if op.asm_start, else one launch with atomic adds
  void launch_cell_diagonal(hipStream_t s, const OperatorData &op, void *diag, const double *a1d,
                            const double *m1d, const uint32_t *lists = nullptr, const uint32_t *list_start = nullptr,
                            int n_lists = 0);
  // transfers
  void launch_prolongate(hipStream_t s, const TransferData &t, void *fine, const void *coarse, bool add,
                         bool with_constraints);
  void launch_restrict_add(hipStream_t s, const TransferData &t, void *coarse, const void *fine,
                           bool with_constraints);

  // ---- vector kernels (mgx_vector.hip) ----
  void launch_copy_cast(hipStream_t s, void *dst, int dn, const void *src, int sn, size_t n);
  void launch_add_cast(hipStream_t s, void *dst, int dn, const void *src, int sn, size_t n);
  void launch_sadd(hipStream_t s, int number, void *x, double sx, double a, const void *v, size_t n);
  // res = rhs - res over [0,n)
  void launch_rhs_minus(hipStream_t s, int number, void *res, const void *rhs, size_t n);
  // dst[c] = src[c] for constrained c  /  res[c] = rhs[c] - lhs[c]
  void launch_constrained_copy(hipStream_t s, int number, void *dst, const void *src, const uint32_t *list,
                               uint32_t count);
  void launch_constrained_residual(hipStream_t s, int number, void *res, const void *rhs, const void *lhs,
                                   const uint32_t *list, uint32_t count);
  void launch_constrained_set(hipStream_t s, int number, void *v, double value, const uint32_t *list,
                              uint32_t count);
  void launch_invert(hipStream_t s, int number, void *v, size_t n);
  void launch_scatter_values(hipStream_t s, int number, void *v, const uint32_t *idx_dev, const double *val_dev,
                             uint32_t count);
  // Chebyshev updates (PreconditionChebyshev internal::vector_updates):
  //   mode 0: x = f2 * dinv * b, x_old = 0
  //   mode 1: x_new = x + f2 * dinv * (b - t)                       (x_old <- x)
  //   mode 2: x_new = x + f1 * (x - x_old) + f2 * dinv * (b - t)     (x_old <- x)
  void launch_cheb_update(hipStream_t s, int number, int mode, void *x, void *x_old, const void *b,
                          const void *t, const void *dinv, double f1, double f2, size_t n);
  void launch_cheb_init(hipStream_t s, int number, void *x, const void *b, const void *dinv, double f2, size_t n);
  // ax == nullptr: (A x)_c = x_c (constrained rows); otherwise the product is read from ax
  void launch_cheb_constrained(hipStream_t s, int number, int mode, const void *x, void *out, const void *b,
                               const void *dinv, double f1, double f2, const uint32_t *list, uint32_t count,
                               const void *ax = nullptr, const void *old = nullptr, double f0 = 0.);
  // interface exchange helpers
  void launch_pack(hipStream_t s, int number, void *buf, const void *v, const uint32_t *list, uint32_t count);
  // DG <-> FE_Q transfer on one mesh (mgx_kernels.hip): to_dg: dg += P cg, else cg += P^T dg
  // eight_colours (restriction only): cells c and c' with c % 8 == c' % 8 share no DoF -- eight launches with
  // plain read-modify-writes instead of one with atomics
  void launch_dg_cg_transfer(hipStream_t s, int number, int p, bool to_dg, void *dst, const void *src,
                             const uint32_t *idx27, uint32_t n_cells, const void *P1, bool eight_colours = false);
  void launch_zero_head_copy_tail(hipStream_t s, int number, void *dst, const void *src, uint32_t n_head, uint32_t n);
  void launch_scatter_map(hipStream_t s, int number, void *dst, const void *src, const uint32_t *map,
                          const uint8_t *mask, uint32_t n);
  void launch_pack_all(hipStream_t s, int number, void *const *send, const uint32_t *start, int n_neighbors,
                       const void *v, const uint32_t *index, const uint8_t *seg, uint32_t total);
  void launch_unpack_ordered(hipStream_t s, int number, void *const *recv, int n_neighbors, void *v,
                             const uint32_t *shared, const uint32_t *csr_start, const uint8_t *csr_k,
                             const uint32_t *csr_pos, uint32_t n_shared);
  // Chebyshev post-operation of the interface DoFs folded into the ordered unpack (one launch instead of two, and the
  // constrained rows -- A x = x -- ride along): what launch_cheb_constrained would do with ax = the completed sums
  struct ChebList
  {
    int             mode;
    const void     *x, *b, *dinv, *old;
    void           *out;
    double          f1, f2, f0;
    const uint32_t *constrained;
    uint32_t        n_constrained;
  };
  void launch_unpack_ordered_cheb(hipStream_t s, int number, void *const *recv, int n_neighbors, void *v,
                                  const uint32_t *shared, const uint32_t *csr_start, const uint8_t *csr_k,
                                  const uint32_t *csr_pos, uint32_t n_shared, const ChebList &post);
  void launch_unpack_add(hipStream_t s, int number, void *v, const void *buf, const uint32_t *list, uint32_t count);
  void launch_list_residual(hipStream_t s, int number, void *res, const void *rhs, const uint32_t *list,
                            uint32_t count); // res[i] = rhs[i] - res[i]
  void launch_dot_list(hipStream_t s, int number, const void *x, const void *y, const uint32_t *list,
                       uint32_t count, double *partial_dev, double *result_dev);
  void launch_index_mod11(hipStream_t s, int number, void *v, const uint32_t *global_index_dev, double mean,
                          size_t n); // v[i] = (gid(i) % 11) - mean  (gid = i if the index array is null)
  // partial[0..n_blocks) block sums of x.y, then reduced into *result_dev (double)
  void launch_dot(hipStream_t s, int number, const void *x, const void *y, size_t n, double *partial_dev,
                  double *result_dev);
  constexpr int kDotBlocks = 1024;
  // CG vector updates with fused reductions
  //   x += alpha d ; r -= alpha h ; result = r.r
  void launch_cg_update(hipStream_t s, int number, void *x, void *r, const void *d, const void *h, double alpha,
                        size_t n, double *partial_dev, double *result_dev);
  //   d = z + beta d
  void launch_xpby(hipStream_t s, int number, void *d, const void *z, double beta, size_t n);
  // mixed-precision PCG: the same with the fp32 copy of the new residual written along / the fp32 vector read directly
  void launch_cg_update_f32copy(hipStream_t s, double *x, double *r, const double *d, const double *h, double alpha, size_t n,
                                float *r32, double *partial_dev, double *result_dev);
  void launch_dot_f64_f32(hipStream_t s, const double *x, const float *y, size_t n, double *partial_dev, double *result_dev);
  void launch_xpby_f64_f32(hipStream_t s, double *d, const float *z, double beta, size_t n);
  // z = dinv .* r ; result = r.z
  void launch_jacobi_dot(hipStream_t s, int number, void *z, const void *dinv, const void *r, size_t n,
                         double *partial_dev, double *result_dev);
  // fused PCG helpers (mgx_vector.hip); the uint32_t results are the numbers of sum quadruples the
  // kernels appended at `partials` (added up by launch_reduce4)
  uint32_t launch_cg_list_update(hipStream_t s, int number, const uint32_t *list, uint32_t count, double alpha,
                                 double beta, const void *r, void *q, void *p, void *x, double *partials);
  // fused transfers on a decomposed level (TransferData::ifr_* / ifp_*, mgx_vector.hip):
  // coarse += R (b - ax) over the owned interface DoFs;  xi = x + P e, out = xi + f2 dinv (b - ax), x = xi on the shared DoFs
  // coarse[d] = ordered sum of the bricks' restricted values (TransferData::coarse_scratch), every coarse DoF written
  void launch_coarse_assemble(hipStream_t s, int number, const TransferData &tr, void *coarse);
  void launch_interface_restrict(hipStream_t s, int number, const TransferData &tr, void *coarse, const void *b, const void *ax);
  void launch_interface_prolong_cheb(hipStream_t s, int number, const TransferData &tr, const uint32_t *shared, const void *e,
                                     void *x, void *out, const void *b, const void *dinv, double f2, const void *ax);
  uint32_t launch_axpy_norm(hipStream_t s, int number, void *r, const void *q, double factor, size_t n, double *partials);
  void     launch_cg_pre(hipStream_t s, int number, void *x, void *p, void *q, double alpha, double beta, size_t n);
  uint32_t launch_dot4(hipStream_t s, int number, const void *q, const void *p, const void *r, size_t n, double *partials);
  void     launch_residual_pre(hipStream_t s, int defect_number, void *defect, const double *residual, const double *update,
                               double factor, size_t n);
  uint32_t launch_residual_post(hipStream_t s, int z_number, const void *z, double *residual, double *update, double factor,
                                size_t n_free, size_t n, double *partials);
} // namespace mgx
