// mgx_api.cpp -- implementation of the C ABI declared in include/mgx.h: host-side control flow of
// LaplaceOperator / PreconditionChebyshev / MGTransferMatrixFree / MultigridSolver on top of the
// HIP kernels.  Reference citations (file:line) are relative to the reference root; deal.II
// semantics follow SURVEY.md 8a rows R, S, T and Appendix D.
#include "../../include/mgx.h"
#include "mgx_bricks.hpp"
#include "mgx_internal.hpp"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h> // types only: the library is bound at run time (dlopen), single-GPU users need no RCCL

#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

using namespace mgx;

namespace
{
  thread_local std::string g_last_error;

  int fail(int code, const std::string &msg)
  {
    g_last_error = msg;
    return code;
  }

#define MGX_HIP(call)                                                                               \
  do                                                                                                \
    {                                                                                               \
      hipError_t e_ = (call);                                                                       \
      if (e_ != hipSuccess)                                                                         \
        return fail(MGX_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_) + " (" + __FILE__ + ":" + \
                                   std::to_string(__LINE__) + ")");                                 \
    }                                                                                               \
  while (0)

#define MGX_TRY(call)     \
  do                      \
    {                     \
      int s_ = (call);    \
      if (s_ != MGX_OK)   \
        return s_;        \
    }                     \
  while (0)

#define MGX_REQUIRE(cond, msg)                        \
  do                                                  \
    {                                                 \
      if (!(cond))                                    \
        return fail(MGX_ERR_INVALID_ARGUMENT, (msg)); \
    }                                                 \
  while (0)

  inline size_t number_size(int number) { return number == MGX_F64 ? 8 : 4; }

  // MGX_TRACE=1 prints the host-side control flow to stderr (debugging aid); set by the first
  // context created with it
  bool g_trace = false;
  bool trace_on() { return g_trace; }
#define MGX_TRACE(...)                  \
  do                                    \
    {                                   \
      if (trace_on())                   \
        {                               \
          std::fprintf(stderr, "[mgx] " __VA_ARGS__); \
          std::fputc('\n', stderr);     \
          std::fflush(stderr);          \
        }                               \
    }                                   \
  while (0)
} // namespace

int mgx::report_error(int code, const char *message) { return fail(code, message ? message : ""); }

// The environment carries the thresholds a deployment may want to tune, nothing else: no variable
// changes what is computed or selects a diagnostic / cross-check code path.  Those are options of
// a context, set explicitly with mgx_context_set_option (tests, A/B timings, tools).
#ifndef MGX_CELLS_FORM
#define MGX_CELLS_FORM 0
#endif
extern "C" int mgx_has_cells_form(void) { return MGX_CELLS_FORM; }

mgx::Tunables mgx::Tunables::from_environment()
{
  Tunables t;
  auto     num = [](const char *name, uint32_t dflt) {
    const char *e = std::getenv(name);
    return e ? (uint32_t)std::strtoul(e, nullptr, 10) : dflt;
  };
  t.trace               = std::getenv("MGX_TRACE") != nullptr;
  t.brick_min           = num("MGX_BRICK_MIN", t.brick_min);
  t.brick_min_from_env  = std::getenv("MGX_BRICK_MIN") != nullptr;
  t.overlap_min         = num("MGX_OVERLAP_MIN_BRICKS", t.overlap_min);
  t.restrict_colour_min = num("MGX_RESTRICT_COLOUR_MIN", t.restrict_colour_min);
  t.cell_colour_min     = num("MGX_CELL_COLOUR_MIN", t.cell_colour_min);
  t.free_max_bricks     = num("MGX_FREE_MAX_BRICKS", t.free_max_bricks);
  t.free_one_max        = num("MGX_FREE_ONE_MAX", t.free_one_max);
  t.graph_max_dofs      = num("MGX_GRAPH_MAX_DOFS", t.graph_max_dofs);
  return t;
}

bool mgx::Tunables::set(const std::string &name, double value)
{
  const bool     on = value != 0.;
  const uint32_t u  = value <= 0. ? 0u : (value >= 4294967295. ? 0xFFFFFFFFu : (uint32_t)value);
  struct Entry
  {
    const char *name;
    bool       *flag;
    uint32_t   *number;
  };
  const Entry table[] = {
    {"trace", &trace, nullptr},
    {"brick_min", nullptr, &brick_min},
    {"overlap_min_bricks", nullptr, &overlap_min},
    {"restrict_colour_min", nullptr, &restrict_colour_min},
    {"cell_colour_min", nullptr, &cell_colour_min},
    {"free_max_bricks", nullptr, &free_max_bricks},
    {"free_one_max", nullptr, &free_one_max},
    {"graph_max_dofs", nullptr, &graph_max_dofs},
    // code-path selectors (numerically equivalent paths; tests compare them)
    {"general_kernel", &general_kernel, nullptr},
    {"no_bricks", &no_bricks, nullptr},
#if MGX_CELLS_FORM // cross-check builds only (make crosscheck)
    {"cells_form", &cells_form, nullptr},
    {"brick_wide_max", nullptr, &wide_max},
#endif
    {"macro_wg_per_cu_x16", nullptr, &macro_wg_x16},
    {"no_diag_table", &no_diag_table, nullptr},
    {"no_macro_v2", &no_macro_v2, nullptr},
    {"no_general_bricks", &no_general_bricks, nullptr},
    {"general_brick_min", nullptr, &general_brick_min},
    {"fused_prolong_min_bricks", nullptr, &fused_prolong_min_bricks},
    {"roctx", &roctx, nullptr},
    {"no_fused_init", &no_fused_init, nullptr},
    {"no_fused_restrict", &no_fused_restrict, nullptr},
    {"no_fused_prolong", &no_fused_prolong, nullptr},
    {"force_fused_transfers", &force_fused_transfers, nullptr},
    {"transfer_v1", &transfer_v1, nullptr},
    {"restrict_atomic", &restrict_atomic, nullptr},
    {"exchange_unfused", &exchange_unfused, nullptr},
    {"no_graph", &no_graph, nullptr},
    {"dg_no_overlap", &dg_no_overlap, nullptr},
    {"dg_unmerged_restrict", &dg_unmerged_restrict, nullptr},
    {"no_fused_decomposed", &no_fused_decomposed, nullptr},
    {"no_fused_assembly", &no_fused_assembly, nullptr},
    {"no_fused_residual", &no_fused_residual, nullptr},
    {"no_restrict_scratch", &no_restrict_scratch, nullptr},
    // emulation of a rank of a decomposed mesh on one GPU (tools/rank_emulation.py): the results are WRONG
    {"rccl_selftest", &rccl_selftest, nullptr},
  };
  for (const Entry &e : table)
    if (name == e.name)
      {
        if (e.flag)
          *e.flag = on;
        else
          *e.number = u;
        if (name == "brick_min")
          brick_min_from_env = true; // taken as given (no p <= 2 doubling)
        return true;
      }
  return false;
}

struct ExchangePlan
{
  std::vector<uint32_t> not_owned_host; // DoFs a lower rank holds as well (host copy of not_owned_dev)
  int                    plan_id = 0, number = MGX_F64, self_pos = 0;
  std::vector<int>       rank;
  std::vector<uint32_t>  count;
  std::vector<uint32_t *> index_dev;
  std::vector<void *>    send, recv;
  std::vector<uint8_t>   owns_buffers;
  uint32_t              *shared_dev = nullptr, *not_owned_dev = nullptr;
  uint32_t               n_shared = 0, n_not_owned = 0;
  void                  *own_buf = nullptr;
  // fused form: one pack launch over the concatenated lists, one ordered unpack launch over the
  // interface DoFs (CSR of their contributions in ascending rank order, 255 = the rank's own sum)
  std::vector<uint32_t>  start;              // [n_neighbors + 1] offsets into the concatenation
  uint32_t              *all_index_dev = nullptr, *csr_start_dev = nullptr, *csr_pos_dev = nullptr;
  uint8_t               *all_seg_dev = nullptr, *csr_k_dev = nullptr;
  bool                   fused = false;
};

// RCCL entry points, bound lazily.  The process may already hold an RCCL (torch bundles one): that
// instance is reused so that only one RCCL runtime is active.
struct RcclApi
{
  void *lib = nullptr;
  decltype(&ncclGetUniqueId)    GetUniqueId    = nullptr;
  decltype(&ncclCommInitRank)   CommInitRank   = nullptr;
  decltype(&ncclCommDestroy)    CommDestroy    = nullptr;
  decltype(&ncclGroupStart)     GroupStart     = nullptr;
  decltype(&ncclGroupEnd)       GroupEnd       = nullptr;
  decltype(&ncclSend)           Send           = nullptr;
  decltype(&ncclRecv)           Recv           = nullptr;
  decltype(&ncclAllReduce)      AllReduce      = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  bool load()
  {
    if (lib)
      return true;
    const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names)
      if ((lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD)))
        break;
    if (!lib)
      for (const char *n : names)
        if ((lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL)))
          break;
    if (!lib)
      return false;
#define MGX_RCCL_SYM(name)                                        \
  name = reinterpret_cast<decltype(name)>(dlsym(lib, "nccl" #name)); \
  if (!name)                                                      \
    {                                                             \
      lib = nullptr;                                              \
      return false;                                               \
    }
    MGX_RCCL_SYM(GetUniqueId)
    MGX_RCCL_SYM(CommInitRank)
    MGX_RCCL_SYM(CommDestroy)
    MGX_RCCL_SYM(GroupStart)
    MGX_RCCL_SYM(GroupEnd)
    MGX_RCCL_SYM(Send)
    MGX_RCCL_SYM(Recv)
    MGX_RCCL_SYM(AllReduce)
    MGX_RCCL_SYM(GetErrorString)
#undef MGX_RCCL_SYM
    return true;
  }
};
static RcclApi &rccl_api()
{
  static RcclApi api;
  return api;
}

struct mgx_context_s
{
  int         device = 0;
  Tunables    tun; // environment switches, read once in mgx_context_create
  hipStream_t stream = nullptr;
  bool        borrowed_stream = false; // the stream of the context of the decomposed hierarchy above (agglomerated levels)
  // interface exchange overlapped with the interior bricks: side stream and the two events that
  // order it against `stream` (created with the communicator)
  hipStream_t side     = nullptr;
  hipEvent_t  ev_iface = nullptr, ev_side = nullptr;
  double     *partial_dev = nullptr; // kDotBlocks block partials
  double     *result_dev  = nullptr; // 4 scalars
  double     *result_host = nullptr; // pinned
  // cell-loop launch profiling (mgx_profile_*)
  bool                                        profile = false;
  struct Bracket
  {
    hipEvent_t start, stop;
    int        form, launches;
  };
  std::vector<Bracket> ev_pool, ev_used;
  // domain decomposition
  bool                                      has_comm = false;
  mgx_comm_desc                             comm{};
  // native transport: RCCL send/recv and allreduce on `stream`, no host synchronisation
  ncclComm_t                                nccl = nullptr;
  bool                                      use_rccl = false;
  int                                       rccl_rank = 0, rccl_size = 1;
  double                                   *ar_dev = nullptr; // allreduce scratch (8 doubles)
  std::vector<std::pair<size_t, struct ExchangePlan *>> plans; // (vector length, plan) for dot products
};

struct mgx_operator_s
{
  mgx_context_t ctx = nullptr;
  OperatorData  d;
  double        S[kMaxN * kMaxN], D[kMaxN * kMaxN], w[kMaxN];
  bool          has_diag = false;
  bool          profiled = false;
  uint32_t     *global_index_dev = nullptr; // optional numbering-independent index (smoother start vector)
  double        start_sum = 0, start_count = 0; // sum of (index mod 11) and number of the owned DoFs
  std::unique_ptr<ExchangePlan> plan;       // interface exchange of a decomposed mesh
  // fused PCG (mgx_vmult_with_cg_update): partial sums, their result, carrier of the brick loop
  double  *cg_partials = nullptr, *cg_result = nullptr;
  void    *cg_carrier  = nullptr;
  bool     constrained_last = false; // the constrained DoFs are exactly [n_dofs - n_constrained, n_dofs)
};

struct mgx_smoother_s
{
  mgx_operator_t    op = nullptr;
  mgx_smoother_info info{};
  void             *x_old = nullptr, *tmp = nullptr;
  void             *x_old2 = nullptr; // third iterate buffer (odd number of fused iterations in step())
  // AdditionalData::PolynomialType::fourth_kind (multigrid_solver.h:951-952): info.delta = lambda_max
  bool              fourth_kind = false;
  double            range_a     = 0; // lower end of the smoothing range (first-kind delta / theta)
  // factor of the first step, and factor1 / factor2 of iteration k = 0, 1, ... of the recurrence
  double first_factor() const { return fourth_kind ? 4. / (3. * info.delta) : 1. / info.theta; }
  void   next_factors(int k, double &rhok, double &f1, double &f2) const
  {
    if (fourth_kind)
      {
        f1 = (2. * k + 1.) / (2. * k + 5.);
        f2 = (8. * k + 12.) / (info.delta * (2. * k + 5.));
        return;
      }
    const double sigma = info.theta / info.delta, rhokp = 1. / (2. * sigma - rhok);
    f1   = rhokp * rhok;
    f2   = 2. * rhokp / info.delta;
    rhok = rhokp;
  }
};

struct mgx_transfer_s
{
  mgx_operator_t coarse = nullptr, fine = nullptr;
  TransferData   d;
  void          *scratch = nullptr; // decomposed mesh: coarse-level scratch of restrict_and_add
};

struct mgx_solver_s
{
  mgx_context_t               ctx = nullptr;
  int                         n_levels = 0, degree = 0, n_cycles = 1, vnumber = MGX_F64;
  std::vector<mgx_operator_t> matrix, matrix_dp;
  std::vector<mgx_transfer_t> transfer, transfer_dp;
  std::vector<mgx_smoother_t> smooth;
  std::vector<double *>       solution, rhs, residual;         // fp64 (multigrid_solver.h:709-719)
  std::vector<void *>         defect, t, solution_update;      // V-cycle precision (:725-735)
  std::vector<uint32_t *>     bc_index_dev;
  std::vector<double *>       bc_value_dev, bc_zero_dev;
  std::vector<uint32_t>       bc_count;
  double                     *cg_r = nullptr, *cg_z = nullptr, *cg_d = nullptr, *cg_h = nullptr;
  bool                        timing = false;
  std::vector<double>         timings; // n_levels*6
  // The V-cycle below `graph_level` is a fixed sequence of small, launch-latency-bound kernels on
  // fixed buffers: it is captured into a HIP graph on its second execution and replayed afterwards
  int             graph_level = -1, graph_calls = 0;
  bool            graph_failed = false;
  hipGraph_t      graph = nullptr;
  hipGraphExec_t  graph_exec = nullptr;
  // Agglomeration of the coarse levels of a decomposed hierarchy (mgx_solver_set_agglomeration):
  // the V-cycle below agg_level runs on agg_solver, an undecomposed copy of those levels that
  // every rank holds on a context of its own (no exchange, graph replay)
  mgx_solver_t    agg_solver = nullptr;
  int             agg_level  = -1;
  int             agg_offset = 0;       // level agg_level of this solver is level agg_level + agg_offset of agg_solver (its finest)
  uint32_t       *agg_map    = nullptr; // device [n_dofs(agg_level)]: local DoF -> DoF of agg_solver's level
  uint8_t        *agg_owned  = nullptr; // device: 1 where this rank owns the DoF
  hipEvent_t      agg_in = nullptr, agg_out = nullptr;
  std::vector<double> agg_host;         // callback transport: staging of the allreduce
  std::vector<double> cg_history;       // residual norms of the last solve_cg: start, then one per iteration
};

namespace
{
  int read_result(mgx_context_t ctx, double *out)
  {
    MGX_HIP(hipMemcpyAsync(ctx->result_host, ctx->result_dev, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    MGX_HIP(hipStreamSynchronize(ctx->stream));
    *out = ctx->result_host[0];
    return MGX_OK;
  }

  // in-place sum of a few host doubles over the ranks
  int comm_allreduce(mgx_context_t ctx, double *values, int count)
  {
    if (!ctx->has_comm)
      return MGX_OK;
    if (ctx->use_rccl)
      {
        if (count > 8)
          return fail(MGX_ERR_INVALID_ARGUMENT, "comm_allreduce: more than 8 values");
        RcclApi &R = rccl_api();
        MGX_HIP(hipMemcpyAsync(ctx->ar_dev, values, sizeof(double) * count, hipMemcpyHostToDevice, ctx->stream));
        if (R.AllReduce(ctx->ar_dev, ctx->ar_dev, (size_t)count, ncclDouble, ncclSum, ctx->nccl, ctx->stream) != ncclSuccess)
          return fail(MGX_ERR_HIP, "ncclAllReduce failed");
        MGX_HIP(hipMemcpyAsync(values, ctx->ar_dev, sizeof(double) * count, hipMemcpyDeviceToHost, ctx->stream));
        MGX_HIP(hipStreamSynchronize(ctx->stream));
        return MGX_OK;
      }
    if (!ctx->comm.allreduce_sum || ctx->comm.allreduce_sum(ctx->comm.user, values, count) != 0)
      return fail(MGX_ERR_HIP, "allreduce_sum callback failed");
    return MGX_OK;
  }

  // x.y over the DoFs this rank owns, summed over the ranks (Vector::operator* / l2_norm with
  // MPI_Allreduce in the reference).  The exchange plan is looked up by the vector length.
  // plan: the ownership plan of the operator the vectors belong to.  nullptr on a decomposed
  // context: looked up by the vector length among the registered operators, which must be
  // unambiguous (the public mgx_dot / mgx_l2_norm take no operator).
  int dot(mgx_context_t ctx, int number, const void *x, const void *y, size_t n, double *out,
          const ExchangePlan *plan = nullptr)
  {
    launch_dot(ctx->stream, number, x, y, n, ctx->partial_dev, ctx->result_dev);
    MGX_TRY(read_result(ctx, out));
    if (!ctx->has_comm)
      return MGX_OK;
    if (!plan)
      {
        for (auto &pr : ctx->plans)
          if (pr.first == n)
            {
              // (operators of one level in two number types share the lists: same sizes)
              if (plan && (plan->n_not_owned != pr.second->n_not_owned || plan->n_shared != pr.second->n_shared))
                return fail(MGX_ERR_INVALID_ARGUMENT, "dot: two operators of this context have vectors of this length but "
                                                      "different ownership; reductions need the operator");
              plan = pr.second;
            }
        if (!plan)
          return fail(MGX_ERR_INVALID_ARGUMENT, "dot: no operator with vectors of this length is registered on this "
                                                "decomposed context (duplicated interface DoFs could not be discounted)");
      }
    if (plan->n_not_owned > 0)
      {
        double dup = 0;
        launch_dot_list(ctx->stream, number, x, y, plan->not_owned_dev, plan->n_not_owned, ctx->partial_dev,
                        ctx->result_dev);
        MGX_TRY(read_result(ctx, &dup));
        *out -= dup;
      }
    return comm_allreduce(ctx, out, 1);
  }

  // Vector::compress(add) for duplicated interface DoFs: every rank ends up with the sum of all
  // sharers' entries, added in ascending rank order on every rank (bitwise identical copies).
  // post (fused exchange plans only; returns with *post_done = true): the Chebyshev update of the interface DoFs and
  // of the constrained rows inside the unpack launch
  int exchange_add(mgx_operator_t op, void *vec, hipStream_t on = nullptr, const ChebList *post = nullptr,
                   bool *post_done = nullptr)
  {
    ExchangePlan *P = op->plan.get();
    if (!P)
      return MGX_OK;
    mgx_context_t ctx = op->ctx;
    hipStream_t   s   = on ? on : ctx->stream;
    const int     num = op->d.number;
    if (P->fused)
      launch_pack_all(s, num, P->send.data(), P->start.data(), (int)P->rank.size(), vec, P->all_index_dev,
                      P->all_seg_dev, P->start.back());
    else
      {
        for (size_t k = 0; k < P->rank.size(); ++k)
          launch_pack(s, num, P->send[k], vec, P->index_dev[k], P->count[k]);
        launch_pack(s, num, P->own_buf, vec, P->shared_dev, P->n_shared);
        launch_constrained_set(s, num, vec, 0.0, P->shared_dev, P->n_shared);
      }
    if (ctx->use_rccl)
      {
        // one group of point-to-point operations on the context's stream: ordered after the pack
        // kernels and before the unpack kernels by the stream itself, the host does not wait
        RcclApi             &R  = rccl_api();
        const ncclDataType_t dt = num == MGX_F64 ? ncclDouble : ncclFloat;
        bool                 ok = R.GroupStart() == ncclSuccess;
        // MGX_RCCL_SELFTEST on a one-rank communicator: every neighbour is the rank itself, so one
        // GPU runs the launch sequence of a rank of a decomposed mesh (tools/rank_emulation.py)
        const bool to_self = ctx->tun.rccl_selftest && ctx->rccl_size == 1;
        for (size_t k = 0; ok && k < P->rank.size(); ++k)
          {
            const int peer = to_self ? 0 : P->rank[k];
            ok = ok && R.Send(P->send[k], P->count[k], dt, peer, ctx->nccl, s) == ncclSuccess;
            ok = ok && R.Recv(P->recv[k], P->count[k], dt, peer, ctx->nccl, s) == ncclSuccess;
          }
        ok = (R.GroupEnd() == ncclSuccess) && ok;
        if (!ok)
          return fail(MGX_ERR_HIP, "RCCL exchange failed");
      }
    else
      {
        MGX_HIP(hipStreamSynchronize(s));
        if (!ctx->comm.exchange ||
            ctx->comm.exchange(ctx->comm.user, P->plan_id, num, (int)P->rank.size(), P->rank.data(),
                               P->count.data(), P->send.data(), P->recv.data()) != 0)
          return fail(MGX_ERR_HIP, "exchange callback failed");
      }
    if (P->fused && post && post_done)
      {
        launch_unpack_ordered_cheb(s, num, P->recv.data(), (int)P->rank.size(), vec, P->shared_dev, P->csr_start_dev,
                                   P->csr_k_dev, P->csr_pos_dev, P->n_shared, *post);
        *post_done = true;
      }
    else if (P->fused)
      launch_unpack_ordered(s, num, P->recv.data(), (int)P->rank.size(), vec, P->shared_dev, P->csr_start_dev,
                            P->csr_k_dev, P->csr_pos_dev, P->n_shared);
    else
      for (size_t k = 0; k <= P->rank.size(); ++k)
        {
          if ((int)k == P->self_pos)
            launch_unpack_add(s, num, vec, P->own_buf, P->shared_dev, P->n_shared);
          if (k < P->rank.size())
            launch_unpack_add(s, num, vec, P->recv[k], P->index_dev[k], P->count[k]);
        }
    MGX_HIP(hipGetLastError());
    return MGX_OK;
  }


  // side stream + events of the overlapped interface exchange (created with the communicator)
  int ensure_side_stream(mgx_context_t ctx)
  {
    if (ctx->side)
      return MGX_OK;
    MGX_HIP(hipSetDevice(ctx->device));
    MGX_HIP(hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking));
    MGX_HIP(hipEventCreateWithFlags(&ctx->ev_iface, hipEventDisableTiming));
    MGX_HIP(hipEventCreateWithFlags(&ctx->ev_side, hipEventDisableTiming));
    return MGX_OK;
  }

  // HIP-event bracket around the cell loop of a profiled operator
  struct ProfileBracket
  {
    mgx_context_t          ctx;
    bool                   on;
    mgx_context_s::Bracket ev;
    ProfileBracket(mgx_operator_t op, int form)
      : ctx(op->ctx)
      , on(op->ctx->profile && op->profiled)
    {
      if (!on)
        return;
      if (ctx->ev_pool.empty())
        {
          if (hipEventCreate(&ev.start) != hipSuccess || hipEventCreate(&ev.stop) != hipSuccess)
            {
              on = false;
              return;
            }
        }
      else
        {
          ev = ctx->ev_pool.back();
          ctx->ev_pool.pop_back();
        }
      ev.form     = form;
      ev.launches = op->d.bricks.available() ? op->d.bricks.n_colours : 1;
      (void)hipEventRecord(ev.start, ctx->stream);
    }
    ~ProfileBracket()
    {
      if (on)
        {
          (void)hipEventRecord(ev.stop, ctx->stream);
          ctx->ev_used.push_back(ev);
        }
    }
  };

  // The brick loop of a level followed, on a decomposed mesh, by the completion of the interface
  // DoFs: exchange of their partial sums in `carrier`, then `fix(stream)` = the list kernel(s) that
  // apply the fused post-operation to them (they were never flagged LAST).
  // Split schedule (BrickData::n_iface_groups > 0): the bricks on the rank interface run first,
  // colour by colour; the exchange and the list kernels go to the side stream behind an event and
  // overlap with the interior bricks, which touch no interface DoF.  (The reference's explicit
  // exchange, laplace_operator_dg.h:986-1057, packs the send-side data first but waits for all
  // requests before its cell loop; deal.II's own loops overlap inside MatrixFree.)  With the callback transport the host blocks in the exchange
  // while the interior launches, enqueued before, execute.
  // Finish(stream, first, count) (colour-free schedule only, `free_schedule`): completes the DoFs
  // [first, first + count) of the operator's list of brick-surface DoFs (launch_surf_finish); the
  // first n_surf_shared of them are the rank-interface DoFs, whose sums go to the carrier.
  // post / post_done: the fix as the post-operation of the unpack launch where the exchange plan allows (exchange_add)
  template <typename Launch, typename Fix, typename Finish>
  int brick_loop_with_exchange(mgx_operator_t op, int form, void *carrier, Launch launch, Fix fix, bool free_schedule,
                               Finish finish, const ChebList *post = nullptr, bool *post_done = nullptr)
  {
    mgx_context_t    ctx = op->ctx;
    hipStream_t      s   = ctx->stream;
    const BrickData &bd  = op->d.bricks;
    const int        ng = free_schedule ? bd.fr.n_groups : bd.n_colours, ni = free_schedule ? bd.fr.n_iface_groups : bd.n_iface_groups;
    const uint32_t   n_sh = bd.fr.n_surf_shared, n_all = bd.fr.n_surf_dofs;
    if (!op->plan || ni == 0 || !ctx->side)
      {
        {
          ProfileBracket pb(op, form);
          launch(s, 0, ng);
          if (free_schedule)
            finish(s, 0u, n_all);
        }
        if (op->plan)
          {
            bool done = false;
            MGX_TRY(exchange_add(op, carrier, nullptr, post, post ? &done : nullptr));
            if (!done)
              fix(s);
            if (post_done)
              *post_done = done;
          }
        return MGX_OK;
      }
    {
      ProfileBracket pb(op, form);
      launch(s, 0, ni);
      if (free_schedule)
        finish(s, 0u, n_sh); // the interface sums are complete after the interface bricks
      MGX_HIP(hipEventRecord(ctx->ev_iface, s));
      MGX_HIP(hipStreamWaitEvent(ctx->side, ctx->ev_iface, 0));
      launch(s, ni, ng); // enqueued before the (possibly host-blocking) exchange
      if (free_schedule)
        finish(s, n_sh, n_all - n_sh);
    }
    {
      bool done = false;
      MGX_TRY(exchange_add(op, carrier, ctx->side, post, post ? &done : nullptr));
      if (!done)
        fix(ctx->side);
      if (post_done)
        *post_done = done;
    }
    MGX_HIP(hipEventRecord(ctx->ev_side, ctx->side));
    MGX_HIP(hipStreamWaitEvent(s, ctx->ev_side, 0));
    return MGX_OK;
  }
  template <typename Launch, typename Fix>
  int brick_loop_with_exchange(mgx_operator_t op, int form, void *carrier, Launch launch, Fix fix)
  {
    return brick_loop_with_exchange(op, form, carrier, launch, fix, false, [](hipStream_t, uint32_t, uint32_t) {});
  }

  // dst = A src on the unconstrained rows (constrained rows untouched).  identity_rows (per-cell
  // kernel only): also dst = src on the constrained rows, set by the launch that zeroes dst when the
  // constrained DoFs are the tail of the vector; *identity_done tells the caller whether it was
  int apply_plain(mgx_operator_t op, void *dst, const void *src, bool identity_rows = false,
                  bool *identity_done = nullptr)
  {
    if (identity_done)
      *identity_done = false;
    hipStream_t s = op->ctx->stream;
    if (op->d.bricks.available())
      {
        const bool fr = op->d.bricks.fr.available();
        return brick_loop_with_exchange(
          op, 0, dst,
          [&](hipStream_t st, int g0, int g1) {
            launch_brick_loop(st, op->d, 0, src, nullptr, nullptr, dst, dst, 0., 0., nullptr, 0., nullptr, nullptr, g0, g1, fr);
          },
          [](hipStream_t) {}, fr,
          [&](hipStream_t st, uint32_t first, uint32_t count) {
            launch_surf_finish(st, op->d, 0, first, count, dst, src, dst, nullptr, nullptr, nullptr, 0., 0., 0.);
          });
      }
    ProfileBracket pb(op, 0);
    if (op->d.gbricks.fr.available() && !op->plan)
      {
        // general operator on its brick schedule: one launch over the bricks, one over the DoFs on brick surfaces
        launch_general_bricks(s, op->d, dst, src);
        launch_surf_finish(s, op->d, 0, 0, op->d.gbricks.fr.n_surf_dofs, dst, src, dst, nullptr, nullptr, nullptr, 0., 0., 0.,
                           nullptr, 0, &op->d.gbricks.fr);
        return MGX_OK;
      }
    if (op->d.asm_start)
      {
        // ordered assembly: the cell loop writes every entry of dst (and the identity rows with it when
        // the constrained DoFs are the tail of the vector)
        const bool tail = identity_rows && identity_done && op->constrained_last;
        launch_cell_loop(s, op->d, dst, src, tail ? src : nullptr, op->d.n_dofs - op->d.n_constrained);
        if (tail)
          *identity_done = true;
        return exchange_add(op, dst);
      }
    // "zero dst within the loop" (laplace_operator.h:590)
    if (identity_rows && identity_done && op->constrained_last && op->d.n_dofs < (1u << 22))
      {
        // small levels are launch-bound: one launch instead of the two fill kernels of a memset
        // plus the copy of the constrained rows
        launch_zero_head_copy_tail(s, op->d.number, dst, src, op->d.n_dofs - op->d.n_constrained, op->d.n_dofs);
        *identity_done = true;
      }
    else
      MGX_HIP(hipMemsetAsync(dst, 0, number_size(op->d.number) * op->d.n_dofs, s));
    launch_cell_loop(s, op->d, dst, src);
    return exchange_add(op, dst); // no-op on a single rank
  }

  // Profiler ranges (context option "roctx"): the reference brackets its phases with LIKWID markers -- fmg_solver,
  // cg_solver, matvec, matvec_sp in the driver (poisson_cube/program.cc:282,309,348,369) and vmult_cheby_<level> around
  // the smoother's operator applications (laplace_operator.h:732-739); the same names go to roctx here, so that a
  // rocprofv3 --marker-trace of a run can be cut by level and phase.  The library is bound at run time.
  struct Roctx
  {
    int (*push)(const char *) = nullptr;
    int (*pop)()              = nullptr;
    Roctx()
    {
      for (const char *n : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"})
        if (void *lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))
          {
            push = reinterpret_cast<int (*)(const char *)>(dlsym(lib, "roctxRangePushA"));
            pop  = reinterpret_cast<int (*)()>(dlsym(lib, "roctxRangePop"));
            if (push && pop)
              return;
          }
      push = nullptr;
      pop  = nullptr;
    }
  };
  Roctx &roctx()
  {
    static Roctx r;
    return r;
  }
  inline void range_push(mgx_context_t ctx, const char *name)
  {
    if (ctx->tun.roctx && roctx().push)
      (void)roctx().push(name);
  }
  inline void range_pop(mgx_context_t ctx)
  {
    if (ctx->tun.roctx && roctx().pop)
      (void)roctx().pop();
  }

  struct Stopwatch
  {
    mgx_solver_t s;
    int          level, slot;
    std::chrono::steady_clock::time_point t0;
    Stopwatch(mgx_solver_t s, int level, int slot)
      : s(s)
      , level(level)
      , slot(slot)
    {
      if (s->ctx->tun.roctx)
        {
          // the phases print_wall_times() reports (multigrid_solver.h:348-371), the smoother under the reference's
          // marker name
          static const char *const names[6] = {"mg_mv", "restrict", "prolongate", "inhomBC", "mg_vec", "vmult_cheby"};
          char                     buf[48];
          std::snprintf(buf, sizeof(buf), "%s_%d", (level == 0 && slot == 0) ? "coarse_solver" : names[slot], level);
          range_push(s->ctx, buf);
        }
      if (s->timing)
        {
          (void)hipStreamSynchronize(s->ctx->stream);
          t0 = std::chrono::steady_clock::now();
        }
    }
    ~Stopwatch()
    {
      if (s->timing)
        {
          (void)hipStreamSynchronize(s->ctx->stream);
          s->timings[6 * level + slot] +=
            std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        }
      range_pop(s->ctx);
    }
  };

  // extreme eigenvalues of a symmetric tridiagonal matrix (Sturm bisection)
  int sturm_count(int n, const double *d, const double *e, double x)
  {
    int    count = 0;
    double q     = d[0] - x;
    if (q < 0)
      ++count;
    for (int i = 1; i < n; ++i)
      {
        if (q == 0)
          q = 1e-300;
        q = d[i] - x - e[i - 1] * e[i - 1] / q;
        if (q < 0)
          ++count;
      }
    return count;
  }

  void tridiag_extreme(int n, const double *d, const double *e, double &lo, double &hi)
  {
    double gl = d[0], gu = d[0];
    for (int i = 0; i < n; ++i)
      {
        const double r = (i > 0 ? std::fabs(e[i - 1]) : 0) + (i < n - 1 ? std::fabs(e[i]) : 0);
        gl             = std::min(gl, d[i] - r);
        gu             = std::max(gu, d[i] + r);
      }
    for (int which = 0; which < 2; ++which)
      {
        const int k = which == 0 ? 1 : n;
        double    a = gl, b = gu;
        for (int it = 0; it < 200; ++it)
          {
            const double m = 0.5 * (a + b);
            if (m == a || m == b)
              break;
            if (sturm_count(n, d, e, m) >= k)
              b = m;
            else
              a = m;
          }
        (which == 0 ? lo : hi) = 0.5 * (a + b);
      }
  }
} // namespace

// Point-to-point exchange of packed device buffers with the context's transport (native RCCL group
// on the stream, or the blocking callback), for the parts of the ABI implemented in other
// translation units (the DG ghost-cell update); allreduce of a few host doubles likewise
int mgx::exchange_buffers(mgx_context_t ctx, int plan_id, int number, int n_neighbors, const int *ranks,
                          const uint32_t *counts, void *const *send, void *const *recv, hipStream_t stream)
{
  MGX_REQUIRE(ctx && ctx->has_comm, "exchange_buffers: no communicator on this context");
  hipStream_t s = stream ? stream : ctx->stream;
  if (ctx->use_rccl)
    {
      RcclApi             &R  = rccl_api();
      const ncclDataType_t dt = number == MGX_F64 ? ncclDouble : ncclFloat;
      const bool           to_self = ctx->tun.rccl_selftest && ctx->rccl_size == 1;
      bool                 ok = R.GroupStart() == ncclSuccess;
      for (int k = 0; ok && k < n_neighbors; ++k)
        {
          const int peer = to_self ? 0 : ranks[k];
          ok = ok && R.Send(send[k], counts[k], dt, peer, ctx->nccl, s) == ncclSuccess;
          ok = ok && R.Recv(recv[k], counts[k], dt, peer, ctx->nccl, s) == ncclSuccess;
        }
      ok = (R.GroupEnd() == ncclSuccess) && ok;
      return ok ? MGX_OK : fail(MGX_ERR_HIP, "RCCL exchange failed");
    }
  MGX_HIP(hipStreamSynchronize(s));
  if (!ctx->comm.exchange || ctx->comm.exchange(ctx->comm.user, plan_id, number, n_neighbors, ranks, counts, send, recv) != 0)
    return fail(MGX_ERR_HIP, "exchange callback failed");
  return MGX_OK;
}

hipStream_t mgx::side_stream_begin(mgx_context_t ctx)
{
  if (!ctx || !ctx->side || hipEventRecord(ctx->ev_iface, ctx->stream) != hipSuccess ||
      hipStreamWaitEvent(ctx->side, ctx->ev_iface, 0) != hipSuccess)
    return nullptr;
  return ctx->side;
}

int mgx::side_stream_end(mgx_context_t ctx)
{
  MGX_REQUIRE(ctx && ctx->side, "side_stream_end: no side stream");
  MGX_HIP(hipEventRecord(ctx->ev_side, ctx->side));
  MGX_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_side, 0));
  return MGX_OK;
}

int mgx::allreduce_sum(mgx_context_t ctx, double *values, int count) { return comm_allreduce(ctx, values, count); }

// x.y over the first n entries of two vectors whose leading part is owned by this rank without
// duplicates (DG vectors: owned cells, then ghost cells), summed over the ranks
int mgx::dot_owned_prefix(mgx_context_t ctx, int number, const void *x, const void *y, size_t n, double *out)
{
  launch_dot(ctx->stream, number, x, y, n, ctx->partial_dev, ctx->result_dev);
  MGX_TRY(read_result(ctx, out));
  return comm_allreduce(ctx, out, 1);
}

bool mgx::context_has_comm(mgx_context_t ctx) { return ctx && ctx->has_comm; }

const mgx::Tunables &mgx::context_tunables(mgx_context_t ctx) { return ctx->tun; }

extern "C" {

const char *mgx_last_error(void) { return g_last_error.c_str(); }
const char *mgx_version(void) { return "mgx 0.1 (gfx950)"; }

int mgx_context_create(mgx_context_t *out, int device)
{
  MGX_REQUIRE(out != nullptr, "mgx_context_create: null output");
  int        count = 0;
  hipError_t e     = hipGetDeviceCount(&count);
  if (e != hipSuccess || count == 0)
    return fail(MGX_ERR_NO_DEVICE, "mgx_context_create: no HIP device available (there is no CPU fallback)");
  MGX_REQUIRE(device >= 0 && device < count, "mgx_context_create: device index out of range");
  MGX_HIP(hipSetDevice(device));
  auto ctx    = new mgx_context_s;
  ctx->device = device;
  ctx->tun    = Tunables::from_environment();
  g_trace     = g_trace || ctx->tun.trace;
  MGX_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
  MGX_HIP(hipMalloc((void **)&ctx->partial_dev, sizeof(double) * kDotBlocks));
  MGX_HIP(hipMalloc((void **)&ctx->result_dev, sizeof(double) * 4));
  MGX_HIP(hipHostMalloc((void **)&ctx->result_host, sizeof(double) * 4));
  *out = ctx;
  return MGX_OK;
}

int mgx_context_set_option(mgx_context_t ctx, const char *name, double value)
{
  MGX_REQUIRE(ctx && name, "mgx_context_set_option: null argument");
  if (!ctx->tun.set(name, value))
    return fail(MGX_ERR_INVALID_ARGUMENT, std::string("mgx_context_set_option: unknown option '") + name + "'");
  g_trace = g_trace || ctx->tun.trace;
  return MGX_OK;
}

int mgx_context_destroy(mgx_context_t ctx)
{
  if (!ctx)
    return MGX_OK;
  if (ctx->borrowed_stream)
    (void)hipDeviceSynchronize(); // the stream's owner may be gone already
  else
    (void)hipStreamSynchronize(ctx->stream);
  (void)hipFree(ctx->partial_dev);
  (void)hipFree(ctx->result_dev);
  (void)hipHostFree(ctx->result_host);
  (void)hipFree(ctx->ar_dev);
  if (ctx->nccl)
    (void)rccl_api().CommDestroy(ctx->nccl);
  for (auto *pool : {&ctx->ev_pool, &ctx->ev_used})
    for (auto &ev : *pool)
      {
        (void)hipEventDestroy(ev.start);
        (void)hipEventDestroy(ev.stop);
      }
  if (ctx->side)
    (void)hipStreamDestroy(ctx->side);
  if (ctx->ev_iface)
    (void)hipEventDestroy(ctx->ev_iface);
  if (ctx->ev_side)
    (void)hipEventDestroy(ctx->ev_side);
  if (!ctx->borrowed_stream)
    (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return MGX_OK;
}

int mgx_sync(mgx_context_t ctx)
{
  MGX_REQUIRE(ctx, "mgx_sync: null context");
  MGX_HIP(hipStreamSynchronize(ctx->stream));
  return MGX_OK;
}

int mgx_device_memory_info(mgx_context_t ctx, size_t *free_bytes, size_t *total_bytes)
{
  MGX_REQUIRE(ctx && free_bytes && total_bytes, "mgx_device_memory_info: null argument");
  MGX_HIP(hipSetDevice(ctx->device));
  MGX_HIP(hipMemGetInfo(free_bytes, total_bytes));
  return MGX_OK;
}

void *mgx_context_stream(mgx_context_t ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int mgx_context_set_comm(mgx_context_t ctx, const mgx_comm_desc *comm)
{
  MGX_REQUIRE(ctx && comm && comm->exchange && comm->allreduce_sum && comm->size >= 1 && comm->rank >= 0 &&
                comm->rank < comm->size,
              "mgx_context_set_comm: bad communicator");
  ctx->comm     = *comm;
  ctx->has_comm = comm->size > 1;
  if (ctx->has_comm)
    MGX_TRY(ensure_side_stream(ctx));
  return MGX_OK;
}

int mgx_rccl_unique_id(void *id128)
{
  MGX_REQUIRE(id128, "mgx_rccl_unique_id: null argument");
  static_assert(sizeof(ncclUniqueId) == MGX_RCCL_ID_BYTES, "ncclUniqueId size");
  RcclApi &R = rccl_api();
  if (!R.load())
    return fail(MGX_ERR_UNSUPPORTED, "mgx_rccl_unique_id: librccl not found");
  ncclUniqueId id;
  if (R.GetUniqueId(&id) != ncclSuccess)
    return fail(MGX_ERR_HIP, "ncclGetUniqueId failed");
  std::memcpy(id128, &id, sizeof(id));
  return MGX_OK;
}

int mgx_context_set_rccl(mgx_context_t ctx, int rank, int size, const void *id128)
{
  MGX_REQUIRE(ctx && id128 && size >= 1 && rank >= 0 && rank < size, "mgx_context_set_rccl: bad argument");
  MGX_REQUIRE(!ctx->nccl, "mgx_context_set_rccl: communicator already set");
  RcclApi &R = rccl_api();
  if (!R.load())
    return fail(MGX_ERR_UNSUPPORTED, "mgx_context_set_rccl: librccl not found");
  MGX_HIP(hipSetDevice(ctx->device));
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof(id));
  const ncclResult_t r = R.CommInitRank(&ctx->nccl, size, id, rank);
  if (r != ncclSuccess)
    {
      ctx->nccl = nullptr;
      return fail(MGX_ERR_HIP, std::string("ncclCommInitRank failed: ") + R.GetErrorString(r));
    }
  MGX_HIP(hipMalloc((void **)&ctx->ar_dev, 8 * sizeof(double)));
  ctx->rccl_rank = rank;
  ctx->rccl_size = size;
  ctx->has_comm  = size > 1 || ctx->tun.rccl_selftest;
  if (ctx->has_comm)
    MGX_TRY(ensure_side_stream(ctx));
  ctx->use_rccl  = true;
  return MGX_OK;
}

int mgx_context_use_rccl(mgx_context_t ctx, int enable)
{
  MGX_REQUIRE(ctx, "mgx_context_use_rccl: null context");
  if (enable)
    MGX_REQUIRE(ctx->nccl, "mgx_context_use_rccl: no RCCL communicator on this context");
  else
    MGX_REQUIRE(ctx->comm.exchange && ctx->comm.allreduce_sum, "mgx_context_use_rccl: no callback transport to fall back to");
  MGX_HIP(hipStreamSynchronize(ctx->stream));
  ctx->use_rccl = enable != 0;
  return MGX_OK;
}

int mgx_copy_device(mgx_context_t ctx, void *dst, const void *src, size_t bytes)
{
  MGX_REQUIRE(ctx && (bytes == 0 || (dst && src)), "mgx_copy_device: null argument");
  if (bytes)
    {
      MGX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
      MGX_HIP(hipStreamSynchronize(ctx->stream));
    }
  return MGX_OK;
}

int mgx_range_push(mgx_context_t ctx, const char *name)
{
  MGX_REQUIRE(ctx && name, "mgx_range_push: null argument");
  range_push(ctx, name);
  return MGX_OK;
}

int mgx_range_pop(mgx_context_t ctx)
{
  MGX_REQUIRE(ctx, "mgx_range_pop: null context");
  range_pop(ctx);
  return MGX_OK;
}

int mgx_profile_enable(mgx_context_t ctx, int enable)
{
  MGX_REQUIRE(ctx, "mgx_profile_enable: null context");
  ctx->profile = enable != 0;
  return MGX_OK;
}

int mgx_profile_read(mgx_context_t ctx, int form, uint64_t *launches, double *total_ms)
{
  MGX_REQUIRE(ctx && launches && total_ms, "mgx_profile_read: null argument");
  MGX_HIP(hipStreamSynchronize(ctx->stream));
  double                              sum = 0;
  uint64_t                            n   = 0;
  std::vector<mgx_context_s::Bracket> keep;
  for (auto &ev : ctx->ev_used)
    {
      if (ev.form != form)
        {
          keep.push_back(ev);
          continue;
        }
      float ms = 0;
      MGX_HIP(hipEventElapsedTime(&ms, ev.start, ev.stop));
      sum += ms;
      n += (uint64_t)ev.launches;
      ctx->ev_pool.push_back(ev);
    }
  ctx->ev_used.swap(keep);
  *launches = n;
  *total_ms = sum;
  return MGX_OK;
}

int mgx_malloc(mgx_context_t ctx, void **dptr, size_t bytes)
{
  MGX_REQUIRE(ctx && dptr, "mgx_malloc: null argument");
  MGX_HIP(hipMalloc(dptr, bytes ? bytes : 8));
  return MGX_OK;
}

int mgx_free(mgx_context_t ctx, void *dptr)
{
  MGX_REQUIRE(ctx, "mgx_free: null context");
  if (dptr)
    {
      MGX_HIP(hipStreamSynchronize(ctx->stream));
      MGX_HIP(hipFree(dptr));
    }
  return MGX_OK;
}

int mgx_upload(mgx_context_t ctx, void *dptr, const void *hptr, size_t bytes)
{
  MGX_REQUIRE(ctx && (bytes == 0 || (dptr && hptr)), "mgx_upload: null argument");
  if (bytes)
    {
      MGX_HIP(hipMemcpyAsync(dptr, hptr, bytes, hipMemcpyHostToDevice, ctx->stream));
      MGX_HIP(hipStreamSynchronize(ctx->stream));
    }
  return MGX_OK;
}

int mgx_download(mgx_context_t ctx, void *hptr, const void *dptr, size_t bytes)
{
  MGX_REQUIRE(ctx && (bytes == 0 || (dptr && hptr)), "mgx_download: null argument");
  if (bytes)
    {
      MGX_HIP(hipMemcpyAsync(hptr, dptr, bytes, hipMemcpyDeviceToHost, ctx->stream));
      MGX_HIP(hipStreamSynchronize(ctx->stream));
    }
  return MGX_OK;
}

int mgx_memset_zero(mgx_context_t ctx, void *dptr, size_t bytes)
{
  MGX_REQUIRE(ctx && (bytes == 0 || dptr), "mgx_memset_zero: null argument");
  if (bytes)
    MGX_HIP(hipMemsetAsync(dptr, 0, bytes, ctx->stream));
  return MGX_OK;
}

int mgx_copy_cast(mgx_context_t ctx, void *dst, int dn, const void *src, int sn, size_t n)
{
  MGX_REQUIRE(ctx && (n == 0 || (dst && src)), "mgx_copy_cast: null argument");
  launch_copy_cast(ctx->stream, dst, dn, src, sn, n);
  return MGX_OK;
}

int mgx_add_cast(mgx_context_t ctx, void *dst, int dn, const void *src, int sn, size_t n)
{
  MGX_REQUIRE(ctx && (n == 0 || (dst && src)), "mgx_add_cast: null argument");
  launch_add_cast(ctx->stream, dst, dn, src, sn, n);
  return MGX_OK;
}

int mgx_sadd(mgx_context_t ctx, int number, void *x, double s, double a, const void *v, size_t n)
{
  MGX_REQUIRE(ctx && (n == 0 || (x && v)), "mgx_sadd: null argument");
  launch_sadd(ctx->stream, number, x, s, a, v, n);
  return MGX_OK;
}

int mgx_dot(mgx_context_t ctx, int number, const void *x, const void *y, size_t n, double *result)
{
  MGX_REQUIRE(ctx && result && (n == 0 || (x && y)), "mgx_dot: null argument");
  return dot(ctx, number, x, y, n, result);
}

int mgx_l2_norm(mgx_context_t ctx, int number, const void *x, size_t n, double *result)
{
  MGX_REQUIRE(ctx && result && (n == 0 || x), "mgx_l2_norm: null argument");
  double s = 0;
  MGX_TRY(dot(ctx, number, x, x, n, &s));
  *result = std::sqrt(s);
  return MGX_OK;
}

/* reductions over the DoFs a rank owns, with the ownership taken from the operator itself (mgx_dot finds
 * it through the vector length) */
int mgx_operator_dot(mgx_operator_t op, const void *x, const void *y, double *result)
{
  MGX_REQUIRE(op && x && y && result, "mgx_operator_dot: null argument");
  return dot(op->ctx, op->d.number, x, y, op->d.n_dofs, result, op->plan.get());
}

int mgx_operator_l2_norm(mgx_operator_t op, const void *x, double *result)
{
  MGX_REQUIRE(op && x && result, "mgx_operator_l2_norm: null argument");
  double s = 0;
  MGX_TRY(dot(op->ctx, op->d.number, x, x, op->d.n_dofs, &s, op->plan.get()));
  *result = std::sqrt(s);
  return MGX_OK;
}

int mgx_set_entries(mgx_context_t ctx, int number, void *v, const uint32_t *idx_host, const double *val_host,
                    uint32_t count)
{
  MGX_REQUIRE(ctx && (count == 0 || (v && idx_host && val_host)), "mgx_set_entries: null argument");
  if (count == 0)
    return MGX_OK;
  uint32_t *idx_dev = nullptr;
  double   *val_dev = nullptr;
  MGX_HIP(hipMalloc((void **)&idx_dev, sizeof(uint32_t) * count));
  MGX_HIP(hipMalloc((void **)&val_dev, sizeof(double) * count));
  MGX_HIP(hipMemcpyAsync(idx_dev, idx_host, sizeof(uint32_t) * count, hipMemcpyHostToDevice, ctx->stream));
  MGX_HIP(hipMemcpyAsync(val_dev, val_host, sizeof(double) * count, hipMemcpyHostToDevice, ctx->stream));
  launch_scatter_values(ctx->stream, number, v, idx_dev, val_dev, count);
  MGX_HIP(hipStreamSynchronize(ctx->stream));
  MGX_HIP(hipFree(idx_dev));
  MGX_HIP(hipFree(val_dev));
  return MGX_OK;
}

/* ------------------------------------------------------------------------------------------
 * LaplaceOperator
 * ------------------------------------------------------------------------------------------ */
int mgx_operator_create(mgx_context_t ctx, const mgx_operator_desc *desc, mgx_operator_t *out)
{
  MGX_REQUIRE(ctx && desc && out, "mgx_operator_create: null argument");
  MGX_REQUIRE(desc->degree >= 1 && desc->degree <= MGX_MAX_DEGREE, "mgx_operator_create: degree must be 1..9");
  MGX_REQUIRE(desc->number == MGX_F32 || desc->number == MGX_F64, "mgx_operator_create: bad number type");
  MGX_REQUIRE(desc->n_cells > 0 && desc->n_dofs > 0, "mgx_operator_create: empty level");
  MGX_REQUIRE(desc->idx27 && desc->shape_values && desc->colloc_grad && desc->qweights,
              "mgx_operator_create: missing table");
  MGX_REQUIRE(desc->n_constrained == 0 || desc->constrained, "mgx_operator_create: missing constrained list");
  const int    p = desc->degree, n = p + 1;
  const size_t n_entries = 27 * (size_t)desc->n_cells;
  // host-side validation of operand shapes before any kernel can touch them
  {
    const uint32_t sizes[3] = {1u, (uint32_t)(p - 1), (uint32_t)((p - 1) * (p - 1))};
    for (int pass = 0; pass < 2; ++pass)
      {
        const uint32_t *tab = pass == 0 ? desc->idx27 : desc->idx27_plain;
        if (!tab)
          continue;
        for (size_t i = 0; i < n_entries; ++i)
          {
            const uint32_t b = tab[i];
            if (b == MGX_INVALID_INDEX)
              continue;
            const int e = (int)(i % 27), cx = e % 3, cy = (e / 3) % 3, cz = e / 9;
            const int kind = (cx == 1) + (cy == 1) + (cz == 1);
            uint32_t  len  = kind == 3 ? sizes[2] * (uint32_t)(p - 1) : sizes[kind];
            if (kind == 0)
              len = 1;
            if ((uint64_t)b + len > desc->n_dofs)
              return fail(MGX_ERR_INVALID_ARGUMENT, "mgx_operator_create: compressed index out of range");
          }
      }
    for (uint32_t i = 0; i < desc->n_constrained; ++i)
      if (desc->constrained[i] >= desc->n_dofs)
        return fail(MGX_ERR_INVALID_ARGUMENT, "mgx_operator_create: constrained index out of range");
  }
  // failures below return through the destroy function: device buffers allocated so far are freed
  std::unique_ptr<mgx_operator_s, int (*)(mgx_operator_t)> op(new mgx_operator_s, mgx_operator_destroy);
  op->ctx          = ctx;
  {
    // (any order inside the list)
    std::vector<uint8_t> seen(desc->n_constrained, 0);
    bool                 last = true;
    for (uint32_t i = 0; i < desc->n_constrained && last; ++i)
      {
        const uint32_t c = desc->constrained[i];
        last             = c >= desc->n_dofs - desc->n_constrained && !seen[c - (desc->n_dofs - desc->n_constrained)];
        if (last)
          seen[c - (desc->n_dofs - desc->n_constrained)] = 1;
      }
    op->constrained_last = last;
  }
  OperatorData &d  = op->d;
  d.p              = p;
  d.number         = desc->number;
  d.n_cells        = desc->n_cells;
  d.n_dofs         = desc->n_dofs;
  d.n_constrained  = desc->n_constrained;
  for (int i = 0; i < 6; ++i)
    d.coef[i] = desc->coef[i];
  std::memcpy(op->S, desc->shape_values, sizeof(double) * n * n);
  std::memcpy(op->D, desc->colloc_grad, sizeof(double) * n * n);
  std::memcpy(op->w, desc->qweights, sizeof(double) * n);
  MGX_HIP(hipSetDevice(ctx->device));
  MGX_HIP(hipMalloc((void **)&d.idx27, sizeof(uint32_t) * n_entries));
  MGX_HIP(hipMemcpy(d.idx27, desc->idx27, sizeof(uint32_t) * n_entries, hipMemcpyHostToDevice));
  if (desc->idx27_plain)
    {
      MGX_HIP(hipMalloc((void **)&d.idx27_plain, sizeof(uint32_t) * n_entries));
      MGX_HIP(hipMemcpy(d.idx27_plain, desc->idx27_plain, sizeof(uint32_t) * n_entries, hipMemcpyHostToDevice));
    }
  MGX_HIP(hipMalloc((void **)&d.constrained, sizeof(uint32_t) * (desc->n_constrained + 1)));
  if (desc->n_constrained)
    MGX_HIP(hipMemcpy(d.constrained, desc->constrained, sizeof(uint32_t) * desc->n_constrained,
                      hipMemcpyHostToDevice));
  // 1D mass and stiffness matrices of the separable form, M = S^T W S, K = S^T D^T W D S
  double M1[kMaxN * kMaxN], K1[kMaxN * kMaxN];
  {
    long double G[kMaxN * kMaxN]; // G = D S: derivative of the nodal basis at the quadrature points
    for (int q = 0; q < n; ++q)
      for (int i = 0; i < n; ++i)
        {
          long double g = 0;
          for (int r = 0; r < n; ++r)
            g += (long double)op->D[q * n + r] * op->S[r * n + i];
          G[q * n + i] = g;
        }
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j)
        {
          long double m = 0, k = 0;
          for (int q = 0; q < n; ++q)
            {
              m += (long double)op->w[q] * op->S[q * n + i] * op->S[q * n + j];
              k += (long double)op->w[q] * G[q * n + i] * G[q * n + j];
            }
          M1[i * n + j] = (double)m;
          K1[i * n + j] = (double)k;
        }
  }
  auto fill_basis = [&](auto &b) {
    using T = std::remove_reference_t<decltype(b.S[0])>;
    for (int i = 0; i < n * n; ++i)
      {
        b.S[i] = (T)op->S[i];
        b.D[i] = (T)op->D[i];
      }
    for (int i = 0; i < n; ++i)
      b.w[i] = (T)op->w[i];
    const int H = n / 2;
    auto      eo = [&](auto &E, const double *A) {
      for (int a = 0; a < H; ++a)
        {
          for (int i = 0; i < H; ++i)
            {
              E.ee[a * H + i] = (T)(0.5 * (A[a * n + i] + A[a * n + n - 1 - i]));
              E.eo[a * H + i] = (T)(0.5 * (A[a * n + i] - A[a * n + n - 1 - i]));
            }
          E.mc[a] = (n % 2) ? (T)A[a * n + H] : (T)0;
        }
      E.mhh = (n % 2) ? (T)A[H * n + H] : (T)0;
    };
    eo(b.mass, M1);
    eo(b.lapl, K1);
  };
  // the separable fast path needs the symmetry A[a][b] = A[n-1-a][n-1-b] of M and K (true for
  // any symmetric node/quadrature set); MGX_GENERAL_KERNEL=1 forces the quadrature-point form
  const Tunables &tun = ctx->tun;
  if (!desc->coef_q)
    {
      // the one coefficient tensor of an affine mesh (xx yy zz xy xz yz) must be positive definite
      const double *c  = desc->coef;
      const double  m2 = c[0] * c[1] - c[3] * c[3];
      const double  m3 = c[0] * (c[1] * c[2] - c[5] * c[5]) - c[3] * (c[3] * c[2] - c[5] * c[4]) +
                        c[4] * (c[3] * c[5] - c[1] * c[4]);
      MGX_REQUIRE(c[0] > 0 && m2 > 0 && m3 > 0, "mgx_operator_create: the coefficient tensor is not positive definite");
    }
  d.full_tensor      = !desc->coef_q && (desc->coef[3] != 0. || desc->coef[4] != 0. || desc->coef[5] != 0.);
  const bool general = d.full_tensor || desc->coef_q; // quadrature-point operation with the full tensor
  d.separable        = !tun.general_kernel && !general;
  d.cells_form       = tun.cells_form;
  d.wide_max         = tun.wide_max;
  d.macro_wg_x16     = tun.macro_wg_x16;
  d.macro_v2         = !tun.no_macro_v2;
  for (int a = 0; a < n && d.separable; ++a)
    for (int bb = 0; bb < n; ++bb)
      if (std::fabs(M1[a * n + bb] - M1[(n - 1 - a) * n + n - 1 - bb]) > 1e-12 ||
          std::fabs(K1[a * n + bb] - K1[(n - 1 - a) * n + n - 1 - bb]) > 1e-10 * std::fabs(K1[0]))
        d.separable = false;
  if (d.number == MGX_F64)
    {
      Basis1D<double> b{};
      fill_basis(b);
      MGX_HIP(hipMalloc(&d.basis, sizeof(b)));
      MGX_HIP(hipMemcpy(d.basis, &b, sizeof(b), hipMemcpyHostToDevice));
    }
  else
    {
      Basis1D<float> b{};
      fill_basis(b);
      MGX_HIP(hipMalloc(&d.basis, sizeof(b)));
      MGX_HIP(hipMemcpy(d.basis, &b, sizeof(b), hipMemcpyHostToDevice));
    }
  MGX_HIP(hipMalloc(&d.inv_diag, number_size(d.number) * d.n_dofs));
  if (general)
    {
      // G = D S for the diagonal of the general cell matrix; the per-point coefficient in the
      // operator's number type
      std::vector<double> G((size_t)n * n);
      for (int q = 0; q < n; ++q)
        for (int i = 0; i < n; ++i)
          {
            double g = 0;
            for (int r = 0; r < n; ++r)
              g += op->D[q * n + r] * op->S[r * n + i];
            G[q * n + i] = g;
          }
      const size_t nq = desc->coef_q ? (size_t)desc->n_cells * 6 * n * n * n : 0;
      if (d.number == MGX_F64)
        {
          MGX_HIP(hipMalloc(&d.grad_1d, sizeof(double) * n * n));
          MGX_HIP(hipMemcpy(d.grad_1d, G.data(), sizeof(double) * n * n, hipMemcpyHostToDevice));
          if (nq)
            {
              MGX_HIP(hipMalloc(&d.coef_q, sizeof(double) * nq));
              MGX_HIP(hipMemcpy(d.coef_q, desc->coef_q, sizeof(double) * nq, hipMemcpyHostToDevice));
            }
        }
      else
        {
          std::vector<float> Gf(G.begin(), G.end());
          MGX_HIP(hipMalloc(&d.grad_1d, sizeof(float) * n * n));
          MGX_HIP(hipMemcpy(d.grad_1d, Gf.data(), sizeof(float) * n * n, hipMemcpyHostToDevice));
          if (nq)
            {
              std::vector<float> cf(desc->coef_q, desc->coef_q + nq);
              MGX_HIP(hipMalloc(&d.coef_q, sizeof(float) * nq));
              MGX_HIP(hipMemcpy(d.coef_q, cf.data(), sizeof(float) * nq, hipMemcpyHostToDevice));
            }
        }
    }
  constexpr size_t kAssemblyMaxEntries = (size_t)1 << 26; // ordered assembly up to this many local values per level
  if (general && (desc->n_cells >= tun.cell_colour_min || (size_t)n * n * n * desc->n_cells > kAssemblyMaxEntries))
    {
      // Cell colouring for the general branch: greedy over the cells in their order, two cells
      // conflict if they share a mesh entity that carries DoFs (its first DoF is the key).  On the
      // structured meshes of the provider this gives the 8 parity classes.  More than 32 colours:
      // keep the single launch with atomics.
      const int             pm1 = p - 1;
      std::vector<uint32_t> used(desc->n_dofs, 0u);
      std::vector<uint8_t>  colour(desc->n_cells, 0);
      uint32_t              count[33] = {0};
      int                   n_colours = 0;
      bool                  ok        = true;
      for (uint32_t c = 0; c < desc->n_cells && ok; ++c)
        {
          const uint32_t *ix   = desc->idx27 + 27 * (size_t)c;
          uint32_t        mask = 0;
          for (int e = 0; e < 27; ++e)
            {
              const int inner = (e % 3 == 1) + ((e / 3) % 3 == 1) + (e / 9 == 1);
              if (inner == 3 || (inner > 0 && pm1 == 0) || ix[e] == MGX_INVALID_INDEX)
                continue;
              mask |= used[ix[e]];
            }
          int col = 0;
          while (col < 32 && (mask >> col) & 1u)
            ++col;
          if (col == 32)
            {
              ok = false;
              break;
            }
          colour[c] = (uint8_t)col;
          ++count[col];
          n_colours = std::max(n_colours, col + 1);
          for (int e = 0; e < 27; ++e)
            {
              const int inner = (e % 3 == 1) + ((e / 3) % 3 == 1) + (e / 9 == 1);
              if (inner == 3 || (inner > 0 && pm1 == 0) || ix[e] == MGX_INVALID_INDEX)
                continue;
              used[ix[e]] |= 1u << col;
            }
        }
      if (ok)
        {
          std::vector<uint32_t> order(desc->n_cells), fill(33, 0);
          d.cell_colour_start[0] = 0;
          for (int k = 0; k < n_colours; ++k)
            d.cell_colour_start[k + 1] = d.cell_colour_start[k] + count[k];
          for (int k = 0; k < n_colours; ++k)
            fill[k] = d.cell_colour_start[k];
          for (uint32_t c = 0; c < desc->n_cells; ++c)
            order[fill[colour[c]]++] = c;
          d.n_cell_colours = n_colours;
          MGX_HIP(hipMalloc((void **)&d.cell_order, sizeof(uint32_t) * (size_t)desc->n_cells));
          MGX_HIP(hipMemcpy(d.cell_order, order.data(), sizeof(uint32_t) * (size_t)desc->n_cells, hipMemcpyHostToDevice));
          MGX_TRACE("operator_create: general branch, %u cells in %d colours", desc->n_cells, n_colours);
        }
    }
  // brick schedule for the atomic-free cell loop (mgx_brick.hip); MGX_NO_BRICKS=1 keeps the
  // per-cell kernel (A/B measurements)
  // (builds without the cell-by-cell cross-check kernels schedule bricks only where the macro-element
  // kernel runs: separable operator, vector below the 4 GB of a buffer descriptor)
  const bool macro_covers = d.separable && (uint64_t)desc->n_dofs * number_size(d.number) < 0xFFFFFFF0ull;
  if (!tun.no_bricks && !general && (MGX_CELLS_FORM ? (p <= 4 || d.separable) : macro_covers))
    {
      BrickHost   bh;
      std::string why;
      const mgx_exchange_desc *ex = desc->exchange;
      // The eight colour launches of a brick loop cost about 80 us however small the level is; below
      // that the per-cell kernel (all cells of the level in one launch, atomic scatter) is faster.
      // Measured cross-over with the macro-element kernel on MI355X (tools/time_matvec.py): p = 4 and
      // p = 8 between 216 and 512 bricks (512: 0.082 vs 0.098 ms, 0.090 vs 0.134 ms), p = 2 between 512
      // and 1728 bricks (512: 0.042 vs 0.028 ms).  MGX_BRICK_MIN overrides the threshold as given.
      const uint32_t brick_min   = tun.brick_min_from_env ? tun.brick_min : (p <= 2 ? 2 * tun.brick_min : tun.brick_min);
      const uint32_t brick_cells = p <= 4 ? 64u : 8u;
      // Decomposed mesh: from this many bricks per rank on, the bricks on the rank interface are
      // launched first and the exchange overlaps with the interior bricks.  The split costs one
      // small (latency-bound) launch per colour; DESIGN.md 6 has the measured break-even.
      const uint32_t overlap_min = tun.overlap_min;
      if (desc->n_dofs >= 0x3FFFFFFFu)
        MGX_TRACE("operator_create: per-cell kernel (%u DoFs do not fit the 30-bit entity index)", desc->n_dofs);
      else if (desc->n_cells / brick_cells < brick_min)
        MGX_TRACE("operator_create: per-cell kernel (%u bricks < %u)", desc->n_cells / brick_cells, brick_min);
      else if (build_bricks(p, desc->n_cells, desc->n_dofs, desc->idx27, desc->idx27_plain, desc->brick_colour,
                            ex ? ex->shared : nullptr, ex ? ex->n_shared : 0,
                            ex && desc->n_cells / brick_cells >= overlap_min, bh, why))
        {
          BrickData &b = d.bricks;
          b.n_bricks   = bh.n_bricks;
          b.n_colours  = bh.n_colours;
          b.n_iface_groups = bh.n_iface_groups;
          b.order      = bh.order;
          for (int c = 0; c <= bh.n_colours; ++c)
            b.colour_start[c] = bh.colour_start[c];
          // device table word: bits 0..29 first DoF, bit 30 FIRST, bit 31 LAST (mgx_brick.hip)
          for (size_t i = 0; i < bh.ent_base.size(); ++i)
            if (bh.ent_base[i] != MGX_INVALID_INDEX)
              bh.ent_base[i] |= (uint32_t)(bh.ent_flags[i] & 3u) << 30;
          MGX_HIP(hipMalloc((void **)&b.ent_base, sizeof(uint32_t) * bh.ent_base.size()));
          MGX_HIP(hipMemcpy(b.ent_base, bh.ent_base.data(), sizeof(uint32_t) * bh.ent_base.size(),
                            hipMemcpyHostToDevice));
          b.ent_flags = nullptr;
          if (d.separable)
            {
              // write-out order of the macro-element kernel (mgx_macro.hip)
              std::vector<uint32_t> map;
              build_item_map(p, map);
              MGX_HIP(hipMalloc((void **)&b.item_map, sizeof(uint32_t) * map.size()));
              MGX_HIP(hipMemcpy(b.item_map, map.data(), sizeof(uint32_t) * map.size(), hipMemcpyHostToDevice));
              {
                // ... and of its second pipeline (mgx_macro2.hip): interior of the brick first
                std::vector<uint32_t> map2;
                build_item_map2(p, map2);
                MGX_HIP(hipMalloc((void **)&b.item_map2, sizeof(uint32_t) * map2.size()));
                MGX_HIP(hipMemcpy(b.item_map2, map2.data(), sizeof(uint32_t) * map2.size(), hipMemcpyHostToDevice));
              }
              // Reduced-colour schedule of the plain / residual / Chebyshev forms (mgx_macro.hip, FREE): one
              // class (one launch per level) on levels with at most free_one_max bricks, two classes up to
              // free_max_bricks
              const int n_classes = b.n_bricks <= tun.free_one_max ? 1 : 2;
              FreeHost  fh;
              if (!MGX_MACRO_PAIRS && !tun.cells_form && b.n_bricks <= tun.free_max_bricks &&
                  (uint64_t)desc->n_dofs * number_size(d.number) < 0xFFFFFFF0ull &&
                  build_free_schedule(p, bh, desc->n_dofs, ex ? ex->shared : nullptr, ex ? ex->n_shared : 0,
                                      bh.n_iface_groups > 0, n_classes, map, fh) &&
                  (size_t)b.n_bricks * fh.n_surf * number_size(d.number) < 0xFFFFFFF0ull && fh.n_groups <= 8)
                {
                  FreeSchedule &fr  = b.fr;
                  fr.n_classes      = n_classes;
                  fr.n_groups       = fh.n_groups;
                  fr.n_iface_groups = fh.n_iface_groups;
                  for (int g = 0; g <= fh.n_groups; ++g)
                    fr.group_start[g] = fh.group_start[g];
                  fr.n_surf        = fh.n_surf;
                  fr.n_surf_dofs   = (uint32_t)fh.surf_dof.size();
                  fr.n_surf_shared = fh.n_surf_shared;
                  fh.surf_dof.push_back(0);
                  fh.surf_pos.push_back(0);
                  auto up = [&](uint32_t *&dev, const std::vector<uint32_t> &v) {
                    if (hipMalloc((void **)&dev, sizeof(uint32_t) * v.size()) != hipSuccess)
                      return false;
                    return hipMemcpy(dev, v.data(), sizeof(uint32_t) * v.size(), hipMemcpyHostToDevice) == hipSuccess;
                  };
                  MGX_REQUIRE(up(fr.surf_off, fh.surf_off) && up(fr.surf_dof, fh.surf_dof) && up(fr.surf_start, fh.surf_start) &&
                                up(fr.surf_pos, fh.surf_pos),
                              "mgx_operator_create: out of device memory (reduced-colour schedule)");
                  MGX_HIP(hipMalloc(&fr.priv, (size_t)b.n_bricks * fh.n_surf * number_size(d.number) + 16));
                  MGX_REQUIRE(up(fr.ent, fh.ent), "mgx_operator_create: out of device memory (reduced-colour schedule)");
                  MGX_TRACE("operator_create: reduced-colour schedule, %d classes, %d groups, %u private values per brick, %u "
                            "private DoFs (%u shared)",
                            n_classes, fr.n_groups, fr.n_surf, fr.n_surf_dofs, fr.n_surf_shared);
                }
            }
          MGX_TRACE("operator_create: %u bricks, %d colours", b.n_bricks, b.n_colours);
        }
      else
        MGX_TRACE("operator_create: per-cell kernel (%s)", why.c_str());
    }
  // Brick form of the general tensor branch (mgx_kernels.hip, brick_general_kernel): p = 4, one rank, vectors and
  // coefficient array below the 4 GB of a 32-bit element offset, cells in bricks (hyper_shell and every structured
  // mapped mesh of mgx_cube): the one-launch schedule of the macro-element kernel -- every entity on a brick surface
  // private -- with its finish kernel.  vmult only; the other forms of a general operator keep the per-cell kernel.
  if (general && p == 4 && !desc->exchange && !tun.no_bricks && !tun.no_general_bricks && desc->n_dofs < 0x3FFFFFFFu &&
      (uint64_t)desc->n_dofs * number_size(d.number) < 0xFFFFFFF0ull &&
      desc->n_cells / 64 >= tun.general_brick_min)
    {
      BrickHost   bh;
      std::string why;
      FreeHost    fh;
      std::vector<uint32_t> map;
      build_item_map(p, map);
      if (build_bricks(p, desc->n_cells, desc->n_dofs, desc->idx27, desc->idx27_plain, nullptr, nullptr, 0, false, bh, why) &&
          build_free_schedule(p, bh, desc->n_dofs, nullptr, 0, false, 1, map, fh) && fh.n_groups == 1 &&
          (size_t)bh.n_bricks * fh.n_surf * number_size(d.number) < 0xFFFFFFF0ull)
        {
          BrickData &g = d.gbricks;
          g.n_bricks   = bh.n_bricks;
          FreeSchedule &fr = g.fr;
          fr.n_classes = 1;
          fr.n_groups  = 1;
          fr.group_start[0] = 0;
          fr.group_start[1] = bh.n_bricks;
          fr.n_surf        = fh.n_surf;
          fr.n_surf_dofs   = (uint32_t)fh.surf_dof.size();
          fr.n_surf_shared = 0;
          fh.surf_dof.push_back(0);
          fh.surf_pos.push_back(0);
          auto up = [&](uint32_t *&dev, const std::vector<uint32_t> &v) {
            if (hipMalloc((void **)&dev, sizeof(uint32_t) * v.size()) != hipSuccess)
              return false;
            return hipMemcpy(dev, v.data(), sizeof(uint32_t) * v.size(), hipMemcpyHostToDevice) == hipSuccess;
          };
          MGX_REQUIRE(up(g.order_dev, bh.order) && up(g.item_map, map) && up(fr.surf_off, fh.surf_off) && up(fr.surf_dof, fh.surf_dof) &&
                        up(fr.surf_start, fh.surf_start) && up(fr.surf_pos, fh.surf_pos) && up(fr.ent, fh.ent),
                      "mgx_operator_create: out of device memory (brick schedule of the general operator)");
          MGX_HIP(hipMalloc(&fr.priv, (size_t)bh.n_bricks * fh.n_surf * number_size(d.number) + 16));
          MGX_TRACE("operator_create: general branch on %u bricks, %u private values per brick, %u private DoFs", bh.n_bricks,
                    fr.n_surf, fr.n_surf_dofs);
        }
      else
        MGX_TRACE("operator_create: general branch without bricks (%s)", why.c_str());
    }
  // Ordered assembly for the per-cell kernels: levels without a brick schedule and without cell
  // colours.  For every DoF the positions (cell (p+1)^3 + local index) of its contributions in
  // ascending cell order, from the compressed index table (read_dof_values_compressed,
  // vector_access_reduced.h:153-229; constrained entities contribute nothing).  Larger levels than
  // kAssemblyMaxEntries keep the one launch with atomic adds (last bits not reproducible).
  {
    const size_t     n3 = (size_t)n * n * n, n_local = n3 * desc->n_cells;
    if (!d.bricks.available() && !d.cell_order && n_local <= kAssemblyMaxEntries)
      {
        std::vector<uint32_t> start((size_t)desc->n_dofs + 1, 0), pos;
        auto for_each_local = [&](auto &&f) {
          for (uint32_t c = 0; c < desc->n_cells; ++c)
            {
              const uint32_t *ix = desc->idx27 + 27 * (size_t)c;
              for (int k = 0; k < n; ++k)
                for (int j = 0; j < n; ++j)
                  {
                    const int       cz = k == 0 ? 0 : (k == p ? 2 : 1), oz = cz == 1 ? k - 1 : 0;
                    const int       cy = j == 0 ? 0 : (j == p ? 2 : 1), oy = cy == 1 ? j - 1 : 0;
                    const uint32_t *e  = ix + 3 * (3 * cz + cy);
                    const uint32_t  off = (uint32_t)((cy == 1 ? p - 1 : 1) * oz + oy);
                    const uint32_t  l0  = (uint32_t)(c * n3 + (size_t)(k * n + j) * n);
                    if (e[0] != MGX_INVALID_INDEX)
                      f(e[0] + off, l0);
                    if (e[1] != MGX_INVALID_INDEX)
                      for (int i = 1; i < p; ++i)
                        f(e[1] + off * (uint32_t)(p - 1) + (uint32_t)(i - 1), l0 + (uint32_t)i);
                    if (e[2] != MGX_INVALID_INDEX)
                      f(e[2] + off, l0 + (uint32_t)p);
                  }
            }
        };
        for_each_local([&](uint32_t dof, uint32_t) { ++start[dof + 1]; });
        for (size_t i = 0; i < desc->n_dofs; ++i)
          start[i + 1] += start[i];
        pos.resize(start.back() + 1);
        std::vector<uint32_t> fill(start.begin(), start.end() - 1);
        for_each_local([&](uint32_t dof, uint32_t l) { pos[fill[dof]++] = l; });
        MGX_HIP(hipMalloc((void **)&d.asm_start, sizeof(uint32_t) * start.size()));
        MGX_HIP(hipMalloc((void **)&d.asm_pos, sizeof(uint32_t) * pos.size()));
        MGX_HIP(hipMalloc(&d.cell_scratch, number_size(d.number) * n_local));
        MGX_HIP(hipMemcpy(d.asm_start, start.data(), sizeof(uint32_t) * start.size(), hipMemcpyHostToDevice));
        MGX_HIP(hipMemcpy(d.asm_pos, pos.data(), sizeof(uint32_t) * pos.size(), hipMemcpyHostToDevice));
        MGX_TRACE("operator_create: ordered assembly of the per-cell kernel (%zu contributions)", pos.size() - 1);
      }
  }
  // smoother start vector statistics over the DoFs this rank owns
  {
    std::vector<uint8_t> skip(desc->n_dofs, 0);
    if (desc->exchange)
      for (uint32_t i = 0; i < desc->exchange->n_not_owned; ++i)
        skip[desc->exchange->not_owned[i]] = 1;
    double sum = 0, cnt = 0;
    for (uint32_t i = 0; i < desc->n_dofs; ++i)
      if (!skip[i])
        {
          sum += (double)((desc->global_index ? desc->global_index[i] : i) % 11u);
          cnt += 1;
        }
    op->start_sum   = sum;
    op->start_count = cnt;
  }
  if (desc->global_index)
    {
      MGX_HIP(hipMalloc((void **)&op->global_index_dev, sizeof(uint32_t) * desc->n_dofs));
      MGX_HIP(hipMemcpy(op->global_index_dev, desc->global_index, sizeof(uint32_t) * desc->n_dofs,
                        hipMemcpyHostToDevice));
    }
  if (desc->exchange)
    {
      const mgx_exchange_desc &e = *desc->exchange;
      MGX_REQUIRE(ctx->has_comm, "mgx_operator_create: exchange plan given but no communicator set on the context");
      MGX_REQUIRE(e.n_neighbors >= 0 && (e.n_neighbors == 0 || (e.neighbor_rank && e.count && e.index)),
                  "mgx_operator_create: incomplete exchange plan");
      auto P     = std::make_unique<ExchangePlan>();
      P->plan_id = e.plan_id;
      P->number  = d.number;
      const size_t es = number_size(d.number);
      const int    my_rank = (ctx->nccl && !ctx->comm.exchange) ? ctx->rccl_rank : ctx->comm.rank;
      // MGX_RCCL_SELFTEST: a one-rank communicator may name itself as neighbour (tools/rccl_selftest.py)
      const bool   selftest = tun.rccl_selftest;
      for (int k = 0; k < e.n_neighbors; ++k)
        {
          MGX_REQUIRE(selftest ||
                        (e.neighbor_rank[k] != my_rank && (k == 0 || e.neighbor_rank[k] > e.neighbor_rank[k - 1])),
                      "mgx_operator_create: neighbour ranks must be ascending and differ from the own rank");
          for (uint32_t i = 0; i < e.count[k]; ++i)
            if (e.index[k][i] >= desc->n_dofs)
              return fail(MGX_ERR_INVALID_ARGUMENT, "mgx_operator_create: exchange index out of range");
          if (e.neighbor_rank[k] < my_rank)
            P->self_pos = k + 1;
          P->rank.push_back(e.neighbor_rank[k]);
          P->count.push_back(e.count[k]);
          uint32_t *idx = nullptr;
          MGX_HIP(hipMalloc((void **)&idx, sizeof(uint32_t) * (e.count[k] + 1)));
          MGX_HIP(hipMemcpy(idx, e.index[k], sizeof(uint32_t) * e.count[k], hipMemcpyHostToDevice));
          P->index_dev.push_back(idx);
          void *sb = e.send_buf ? e.send_buf[k] : nullptr, *rb = e.recv_buf ? e.recv_buf[k] : nullptr;
          bool own = !(sb && rb);
          if (own && ctx->comm.alloc_device)
            {
              sb  = ctx->comm.alloc_device(ctx->comm.user, es * (e.count[k] + 1));
              rb  = ctx->comm.alloc_device(ctx->comm.user, es * (e.count[k] + 1));
              own = false;
              MGX_REQUIRE(sb && rb, "mgx_operator_create: the communicator's alloc_device failed");
            }
          if (own)
            {
              MGX_HIP(hipMalloc(&sb, es * (e.count[k] + 1)));
              MGX_HIP(hipMalloc(&rb, es * (e.count[k] + 1)));
            }
          P->send.push_back(sb);
          P->recv.push_back(rb);
          P->owns_buffers.push_back(own ? 1 : 0);
        }
      for (uint32_t i = 0; i < e.n_shared; ++i)
        if (e.shared[i] >= desc->n_dofs)
          return fail(MGX_ERR_INVALID_ARGUMENT, "mgx_operator_create: shared index out of range");
      {
        // Dirichlet DoFs are never exchanged: their rows are the identity on every rank, and the
        // interface post-operations assume the two lists to be disjoint (mgx.h, mgx_exchange_desc)
        std::vector<uint8_t> is_constrained(desc->n_dofs, 0);
        for (uint32_t i = 0; i < desc->n_constrained; ++i)
          is_constrained[desc->constrained[i]] = 1;
        for (uint32_t i = 0; i < e.n_shared; ++i)
          if (is_constrained[e.shared[i]])
            return fail(MGX_ERR_INVALID_ARGUMENT, "mgx_operator_create: a constrained DoF is listed as shared; leave "
                                                  "Dirichlet DoFs out of the exchange plan");
        for (int k = 0; k < e.n_neighbors; ++k)
          for (uint32_t i = 0; i < e.count[k]; ++i)
            if (is_constrained[e.index[k][i]])
              return fail(MGX_ERR_INVALID_ARGUMENT, "mgx_operator_create: a constrained DoF is listed for exchange; "
                                                    "leave Dirichlet DoFs out of the exchange plan");
      }
      P->n_shared    = e.n_shared;
      P->n_not_owned = e.n_not_owned;
      P->not_owned_host.assign(e.not_owned, e.not_owned + e.n_not_owned);
      MGX_HIP(hipMalloc((void **)&P->shared_dev, sizeof(uint32_t) * (e.n_shared + 1)));
      MGX_HIP(hipMalloc((void **)&P->not_owned_dev, sizeof(uint32_t) * (e.n_not_owned + 1)));
      MGX_HIP(hipMalloc(&P->own_buf, es * (e.n_shared + 1)));
      if (e.n_shared)
        MGX_HIP(hipMemcpy(P->shared_dev, e.shared, sizeof(uint32_t) * e.n_shared, hipMemcpyHostToDevice));
      if (e.n_not_owned)
        MGX_HIP(hipMemcpy(P->not_owned_dev, e.not_owned, sizeof(uint32_t) * e.n_not_owned, hipMemcpyHostToDevice));
      // fused pack / ordered unpack tables
      if (e.n_neighbors <= 32 && e.n_neighbors < 255 && !tun.exchange_unfused)
        {
          P->start.assign(e.n_neighbors + 1, 0);
          for (int k = 0; k < e.n_neighbors; ++k)
            P->start[k + 1] = P->start[k] + e.count[k];
          const uint32_t        total = P->start.back();
          std::vector<uint32_t> all_index(total + 1, 0);
          std::vector<uint8_t>  all_seg(total + 1, 0);
          for (int k = 0; k < e.n_neighbors; ++k)
            for (uint32_t i = 0; i < e.count[k]; ++i)
              {
                all_index[P->start[k] + i] = e.index[k][i];
                all_seg[P->start[k] + i]   = (uint8_t)k;
              }
          // contributions per interface DoF in ascending rank order, own sum at self_pos
          std::vector<uint32_t> slot(desc->n_dofs, MGX_INVALID_INDEX);
          for (uint32_t j = 0; j < e.n_shared; ++j)
            slot[e.shared[j]] = j;
          std::vector<uint32_t> cnt(e.n_shared + 1, 0);
          bool                  ok = true;
          for (int k = 0; k < e.n_neighbors && ok; ++k)
            for (uint32_t i = 0; i < e.count[k]; ++i)
              {
                const uint32_t j = slot[e.index[k][i]];
                if (j == MGX_INVALID_INDEX)
                  {
                    ok = false; // an exchanged DoF that is not in the shared list: keep the plain form
                    break;
                  }
                cnt[j]++;
              }
          if (ok)
            {
              std::vector<uint32_t> cs(e.n_shared + 1, 0);
              for (uint32_t j = 0; j < e.n_shared; ++j)
                cs[j + 1] = cs[j] + cnt[j] + 1;
              std::vector<uint8_t>  ck(cs.back() + 1, 0);
              std::vector<uint32_t> cp(cs.back() + 1, 0), fill(cs.begin(), cs.end() - 1);
              for (int k = 0; k <= e.n_neighbors; ++k)
                {
                  if (k == P->self_pos)
                    for (uint32_t j = 0; j < e.n_shared; ++j)
                      ck[fill[j]++] = 255;
                  if (k < e.n_neighbors)
                    for (uint32_t i = 0; i < e.count[k]; ++i)
                      {
                        const uint32_t j = slot[e.index[k][i]];
                        ck[fill[j]]   = (uint8_t)k;
                        cp[fill[j]++] = i;
                      }
                }
              MGX_HIP(hipMalloc((void **)&P->all_index_dev, sizeof(uint32_t) * all_index.size()));
              MGX_HIP(hipMalloc((void **)&P->all_seg_dev, all_seg.size()));
              MGX_HIP(hipMalloc((void **)&P->csr_start_dev, sizeof(uint32_t) * cs.size()));
              MGX_HIP(hipMalloc((void **)&P->csr_k_dev, ck.size()));
              MGX_HIP(hipMalloc((void **)&P->csr_pos_dev, sizeof(uint32_t) * cp.size()));
              MGX_HIP(hipMemcpy(P->all_index_dev, all_index.data(), sizeof(uint32_t) * all_index.size(), hipMemcpyHostToDevice));
              MGX_HIP(hipMemcpy(P->all_seg_dev, all_seg.data(), all_seg.size(), hipMemcpyHostToDevice));
              MGX_HIP(hipMemcpy(P->csr_start_dev, cs.data(), sizeof(uint32_t) * cs.size(), hipMemcpyHostToDevice));
              MGX_HIP(hipMemcpy(P->csr_k_dev, ck.data(), ck.size(), hipMemcpyHostToDevice));
              MGX_HIP(hipMemcpy(P->csr_pos_dev, cp.data(), sizeof(uint32_t) * cp.size(), hipMemcpyHostToDevice));
              P->fused = true;
            }
        }
      ctx->plans.push_back({(size_t)desc->n_dofs, P.get()});
      op->plan = std::move(P);
    }
  *out = op.release();
  return MGX_OK;
}

int mgx_operator_exchange_buffers(mgx_operator_t op, int k, void **send, void **recv, uint32_t *count, int *rank)
{
  MGX_REQUIRE(op && op->plan && k >= 0 && k < (int)op->plan->rank.size(), "mgx_operator_exchange_buffers: bad argument");
  if (send)
    *send = op->plan->send[k];
  if (recv)
    *recv = op->plan->recv[k];
  if (count)
    *count = op->plan->count[k];
  if (rank)
    *rank = op->plan->rank[k];
  return MGX_OK;
}

int mgx_exchange_add(mgx_operator_t op, void *vec)
{
  MGX_REQUIRE(op && vec, "mgx_exchange_add: null argument");
  return exchange_add(op, vec);
}

int mgx_operator_destroy(mgx_operator_t op)
{
  if (!op)
    return MGX_OK;
  (void)hipStreamSynchronize(op->ctx->stream);
  (void)hipFree(op->d.idx27);
  (void)hipFree(op->d.idx27_plain);
  (void)hipFree(op->d.constrained);
  (void)hipFree(op->d.basis);
  (void)hipFree(op->d.inv_diag);
  (void)hipFree(op->d.coef_q);
  (void)hipFree(op->d.grad_1d);
  (void)hipFree(op->d.cell_order);
  (void)hipFree(op->d.asm_start);
  (void)hipFree(op->d.asm_pos);
  (void)hipFree(op->d.cell_scratch);
  (void)hipFree(op->d.bricks.ent_base);
  (void)hipFree(op->d.bricks.ent_flags);
  (void)hipFree(op->d.bricks.item_map);
  (void)hipFree(op->d.bricks.item_map2);
  (void)hipFree(op->d.bricks.fr.ent);
  (void)hipFree(op->d.bricks.fr.surf_off);
  (void)hipFree(op->d.bricks.fr.priv);
  (void)hipFree(op->d.bricks.fr.surf_dof);
  (void)hipFree(op->d.bricks.fr.surf_start);
  (void)hipFree(op->d.bricks.fr.surf_pos);
  (void)hipFree(op->d.gbricks.item_map);
  (void)hipFree(op->d.gbricks.order_dev);
  (void)hipFree(op->d.gbricks.fr.ent);
  (void)hipFree(op->d.gbricks.fr.surf_off);
  (void)hipFree(op->d.gbricks.fr.priv);
  (void)hipFree(op->d.gbricks.fr.surf_dof);
  (void)hipFree(op->d.gbricks.fr.surf_start);
  (void)hipFree(op->d.gbricks.fr.surf_pos);
  (void)hipFree(op->d.diag_items);
  (void)hipFree(op->d.diag_items2);
  (void)hipFree(op->cg_partials);
  (void)hipFree(op->cg_result);
  (void)hipFree(op->cg_carrier);
  (void)hipFree(op->global_index_dev);
  if (op->plan)
    {
      ExchangePlan *P = op->plan.get();
      auto         &v = op->ctx->plans;
      v.erase(std::remove_if(v.begin(), v.end(), [P](const std::pair<size_t, ExchangePlan *> &x) { return x.second == P; }),
              v.end());
      for (size_t k = 0; k < P->rank.size(); ++k)
        {
          (void)hipFree(P->index_dev[k]);
          if (P->owns_buffers[k])
            {
              (void)hipFree(P->send[k]);
              (void)hipFree(P->recv[k]);
            }
        }
      (void)hipFree(P->shared_dev);
      (void)hipFree(P->not_owned_dev);
      (void)hipFree(P->own_buf);
      (void)hipFree(P->all_index_dev);
      (void)hipFree(P->all_seg_dev);
      (void)hipFree(P->csr_start_dev);
      (void)hipFree(P->csr_k_dev);
      (void)hipFree(P->csr_pos_dev);
    }
  delete op;
  return MGX_OK;
}

uint32_t mgx_operator_n_dofs(mgx_operator_t op) { return op ? op->d.n_dofs : 0; }

int mgx_operator_set_profiled(mgx_operator_t op, int profiled)
{
  MGX_REQUIRE(op, "mgx_operator_set_profiled: null operator");
  op->profiled = profiled != 0;
  return MGX_OK;
}
int      mgx_operator_number(mgx_operator_t op) { return op ? op->d.number : -1; }

int mgx_vmult(mgx_operator_t op, void *dst, const void *src)
{
  MGX_REQUIRE(op && dst && src, "mgx_vmult: null argument");
  MGX_REQUIRE(dst != src, "mgx_vmult: dst and src must not alias (laplace_operator.h:573-601)");
  hipStream_t s = op->ctx->stream;
  bool identity_done = false;
  MGX_TRY(apply_plain(op, dst, src, true, &identity_done));
  // dst[c] = src[c] on constrained rows (:592-593)
  if (!identity_done)
    launch_constrained_copy(s, op->d.number, dst, src, op->d.constrained, op->d.n_constrained);
  MGX_HIP(hipGetLastError());
  return MGX_OK;
}

int mgx_vmult_residual(mgx_operator_t op, const void *rhs, const void *lhs, void *res)
{
  MGX_REQUIRE(op && rhs && lhs && res, "mgx_vmult_residual: null argument");
  MGX_REQUIRE(res != lhs && res != rhs, "mgx_vmult_residual: residual must not alias rhs/lhs");
  hipStream_t s = op->ctx->stream;
  if (op->d.bricks.available())
    {
      // zeroing (:617-623) and rhs - A lhs (:624-631) are fused into the brick loop; interface
      // DoFs hold partial sums of A lhs: complete them, then rhs - (.)
      const bool fr = op->d.bricks.fr.available();
      MGX_TRY(brick_loop_with_exchange(
        op, 1, res,
        [&](hipStream_t st, int g0, int g1) {
          launch_brick_loop(st, op->d, 1, lhs, rhs, nullptr, res, res, 0., 0., nullptr, 0., nullptr, nullptr, g0, g1, fr);
        },
        [&](hipStream_t st) {
          launch_list_residual(st, op->d.number, res, rhs, op->plan->shared_dev, op->plan->n_shared);
        },
        fr,
        [&](hipStream_t st, uint32_t first, uint32_t count) {
          launch_surf_finish(st, op->d, 1, first, count, res, lhs, res, rhs, nullptr, nullptr, 0., 0., 0.);
        }));
    }
  else
    {
      MGX_TRY(apply_plain(op, res, lhs));
      launch_rhs_minus(s, op->d.number, res, rhs, op->d.n_dofs); // :624-631
    }
  // res[c] -= lhs[c] on constrained rows (:632-633); the loop never touches them
  launch_constrained_residual(s, op->d.number, res, rhs, lhs, op->d.constrained, op->d.n_constrained);
  MGX_HIP(hipGetLastError());
  return MGX_OK;
}

/* LaplaceOperator::vmult_with_cg_update (laplace_operator.h:638-719) */
int mgx_vmult_with_cg_update(mgx_operator_t op, double alpha, double beta, const void *r, void *q, void *p, void *x,
                             void *scratch, double sums[4])
{
  MGX_REQUIRE(op && r && q && p && x && sums, "mgx_vmult_with_cg_update: null argument");
  MGX_REQUIRE(q != p && q != x && p != x && r != q && r != p && r != x, "mgx_vmult_with_cg_update: vectors must not alias");
  mgx_context_t ctx = op->ctx;
  hipStream_t   s   = ctx->stream;
  const int     num = op->d.number;
  const size_t  n   = op->d.n_dofs;
  constexpr uint32_t kCapacity = 1u << 16; // quadruples
  if (!op->cg_partials)
    {
      MGX_HIP(hipMalloc((void **)&op->cg_partials, sizeof(double) * 4 * kCapacity));
      MGX_HIP(hipMalloc((void **)&op->cg_result, sizeof(double) * 4));
    }
  bool fused = op->d.bricks.available() && op->d.separable && op->d.bricks.item_map && !op->plan;
  if (fused)
    {
      // the brick loop reads q (the preconditioned residual) while neighbouring bricks already hold
      // partial sums of the new q = A p: those travel in a separate carrier vector
      void *carrier = scratch;
      if (!carrier)
        {
          if (!op->cg_carrier)
            MGX_HIP(hipMalloc(&op->cg_carrier, number_size(num) * n));
          carrier = op->cg_carrier;
        }
      uint32_t used = 0;
      {
        ProfileBracket pb(op, 8);
        fused = num == MGX_F64 ? launch_macro_cg_update_f64(s, op->d, alpha, beta, r, q, p, x, carrier, op->cg_partials,
                                                            kCapacity - 2048, &used)
                               : launch_macro_cg_update_f32(s, op->d, alpha, beta, r, q, p, x, carrier, op->cg_partials,
                                                            kCapacity - 2048, &used);
      }
      if (fused)
        {
          // constrained rows: the vector updates of the before-loop hook; the cell loop leaves q = 0 there
          used += launch_cg_list_update(s, num, op->d.constrained, op->d.n_constrained, alpha, beta, r, q, p, x,
                                        op->cg_partials + 4 * (size_t)used);
          launch_reduce4(s, op->cg_partials, used, nullptr, op->cg_result);
          MGX_HIP(hipMemcpyAsync(sums, op->cg_result, 4 * sizeof(double), hipMemcpyDeviceToHost, s));
          MGX_HIP(hipStreamSynchronize(s));
          MGX_HIP(hipGetLastError());
          return MGX_OK;
        }
    }
  // levels without the fused kernel / decomposed meshes: the same operations one after the other
  launch_cg_pre(s, num, x, p, q, alpha, beta, n);
  MGX_TRY(apply_plain(op, q, p)); // constrained rows stay zero, as in the reference's cell loop
  if (!ctx->has_comm)
    {
      const uint32_t used = launch_dot4(s, num, q, p, r, n, op->cg_partials);
      launch_reduce4(s, op->cg_partials, used, nullptr, op->cg_result);
      MGX_HIP(hipMemcpyAsync(sums, op->cg_result, 4 * sizeof(double), hipMemcpyDeviceToHost, s));
      MGX_HIP(hipStreamSynchronize(s));
    }
  else
    {
      MGX_TRY(dot(ctx, num, q, p, n, &sums[0], op->plan.get()));
      MGX_TRY(dot(ctx, num, r, r, n, &sums[1], op->plan.get()));
      MGX_TRY(dot(ctx, num, q, r, n, &sums[2], op->plan.get()));
      MGX_TRY(dot(ctx, num, q, q, n, &sums[3], op->plan.get()));
    }
  MGX_HIP(hipGetLastError());
  return MGX_OK;
}

// Cells of a brick level in launches whose cells share no DoF: brick colour by brick colour (bricks of one launch
// group share no DoF), inside a brick the cells with the same position m mod 8 in Morton order -- the same child of
// every parent -- do not touch.  64 lists; plain adds in an order that does not depend on the run.
static void brick_cell_lists(mgx_operator_t op, std::vector<uint32_t> &lists, std::vector<uint32_t> &list_start)
{
  const BrickData &bd = op->d.bricks;
  const uint32_t   cb = op->d.n_cells / bd.n_bricks; // cells per brick: 64 or 8
  lists.clear();
  list_start.assign(1, 0);
  lists.reserve(op->d.n_cells);
  for (int g = 0; g < bd.n_colours; ++g)
    for (uint32_t q = 0; q < 8; ++q)
      {
        for (uint32_t pos = bd.colour_start[g]; pos < bd.colour_start[g + 1]; ++pos)
          for (uint32_t m = q; m < cb; m += 8)
            lists.push_back(bd.order[pos] * cb + m);
        list_start.push_back((uint32_t)lists.size());
      }
}

int mgx_compute_residual(mgx_operator_t op, void *dst, const void *src, const void *rhs_q)
{
  MGX_REQUIRE(op && dst, "mgx_compute_residual: null argument");
  MGX_REQUIRE(dst != src, "mgx_compute_residual: dst and src must not alias");
  // (the boundary values are gathered through the table that also names the constrained DoFs)
  MGX_REQUIRE(op->d.idx27_plain, "mgx_compute_residual: the operator was created without idx27_plain");
  hipStream_t  s     = op->ctx->stream;
  const size_t bytes = number_size(op->d.number) * op->d.n_dofs;
  // temporaries of this call, released on every path out of it (after the stream has drained)
  struct Scratch
  {
    hipStream_t s;
    void       *zero      = nullptr;
    uint32_t   *lists_dev = nullptr;
    ~Scratch()
    {
      if (zero || lists_dev)
        (void)hipStreamSynchronize(s);
      (void)hipFree(zero);
      (void)hipFree(lists_dev);
    }
  } tmp{s};
  if (!src) // homogeneous boundary values
    {
      MGX_HIP(hipMalloc(&tmp.zero, bytes));
      MGX_HIP(hipMemsetAsync(tmp.zero, 0, bytes, s));
      src = tmp.zero;
    }
  MGX_HIP(hipMemsetAsync(dst, 0, bytes, s));
  // assembly without atomics, as for the diagonal (mgx_compute_diagonal)
  if (op->d.bricks.available())
    {
      std::vector<uint32_t> lists, list_start;
      brick_cell_lists(op, lists, list_start);
      MGX_HIP(hipMalloc((void **)&tmp.lists_dev, sizeof(uint32_t) * (lists.size() + 1)));
      MGX_HIP(hipMemcpyAsync(tmp.lists_dev, lists.data(), sizeof(uint32_t) * lists.size(), hipMemcpyHostToDevice, s));
      launch_cell_residual(s, op->d, dst, src, rhs_q, tmp.lists_dev, list_start.data(), (int)list_start.size() - 1);
      MGX_HIP(hipStreamSynchronize(s)); // (`lists` is pageable host memory in flight until here)
    }
  else if (op->d.cell_order && !op->d.asm_start)
    launch_cell_residual(s, op->d, dst, src, rhs_q, op->d.cell_order, op->d.cell_colour_start, op->d.n_cell_colours);
  else
    launch_cell_residual(s, op->d, dst, src, rhs_q);
  MGX_HIP(hipGetLastError());
  return exchange_add(op, dst); // dst.compress(add), laplace_operator.h:843
}

int mgx_compute_diagonal(mgx_operator_t op)
{
  MGX_REQUIRE(op, "mgx_compute_diagonal: null argument");
  hipStream_t s = op->ctx->stream;
  const int   n = op->d.p + 1;
  // 1D diagonal factors: G = D*S is the gradient of the nodal basis at the quadrature points
  double a1d[kMaxN], m1d[kMaxN];
  for (int i = 0; i < n; ++i)
    {
      double a = 0, m = 0;
      for (int q = 0; q < n; ++q)
        {
          double g = 0;
          for (int r = 0; r < n; ++r)
            g += op->D[q * n + r] * op->S[r * n + i];
          a += op->w[q] * g * g;
          m += op->w[q] * op->S[q * n + i] * op->S[q * n + i];
        }
      a1d[i] = a;
      m1d[i] = m;
    }
  MGX_HIP(hipMemsetAsync(op->d.inv_diag, 0, number_size(op->d.number) * op->d.n_dofs, s));
  // The cell contributions are added up without atomics, in an order that does not depend on the run:
  // brick levels colour by colour (bricks of one launch group share no DoF; inside a brick the cells
  // with the same position m mod 8 in Morton order -- the same child of every parent -- do not touch),
  // cell-coloured levels colour by colour, the others through the ordered assembly.
  if (op->d.bricks.available())
    {
      std::vector<uint32_t> lists, list_start;
      brick_cell_lists(op, lists, list_start);
      uint32_t *lists_dev = nullptr;
      MGX_HIP(hipMalloc((void **)&lists_dev, sizeof(uint32_t) * (lists.size() + 1)));
      MGX_HIP(hipMemcpyAsync(lists_dev, lists.data(), sizeof(uint32_t) * lists.size(), hipMemcpyHostToDevice, s));
      launch_cell_diagonal(s, op->d, op->d.inv_diag, a1d, m1d, lists_dev, list_start.data(), (int)list_start.size() - 1);
      MGX_HIP(hipStreamSynchronize(s));
      MGX_HIP(hipFree(lists_dev));
    }
  else if (op->d.cell_order)
    launch_cell_diagonal(s, op->d, op->d.inv_diag, a1d, m1d, op->d.cell_order, op->d.cell_colour_start, op->d.n_cell_colours);
  else
    launch_cell_diagonal(s, op->d, op->d.inv_diag, a1d, m1d);
  MGX_TRY(exchange_add(op, op->d.inv_diag)); // Vector::compress(add) of the reference's cell_loop
  // set_constrained_entries_to_one + invert (laplace_operator.h:757-765)
  launch_constrained_set(s, op->d.number, op->d.inv_diag, 1.0, op->d.constrained, op->d.n_constrained);
  launch_invert(s, op->d.number, op->d.inv_diag, op->d.n_dofs);
  MGX_HIP(hipGetLastError());
  op->has_diag = true;
  // Is the diagonal the same for every brick (uniform mesh)?  Then the macro-element kernel keeps
  // it in registers instead of streaming it (mgx_macro.hip, DTAB).  MGX_NO_DIAG_TABLE=1: A/B timing.
  for (int which = 0; which < 2; ++which) // the table in the item order of either pipeline of the macro-element kernel
    {
      void          *&slot = which == 0 ? op->d.diag_items : op->d.diag_items2;
      const uint32_t *map  = which == 0 ? op->d.bricks.item_map : op->d.bricks.item_map2;
      if (slot)
        {
          MGX_HIP(hipFree(slot));
          slot = nullptr;
        }
      if (!map || !op->d.separable || op->ctx->tun.no_diag_table)
        continue;
      const uint32_t nb = op->d.p <= 4 ? 4 : 2, g = nb * op->d.p + 1, npts = g * g * g;
      void          *table = nullptr;
      uint32_t      *flag  = nullptr, mismatch = 1;
      MGX_HIP(hipMalloc(&table, number_size(op->d.number) * npts));
      MGX_HIP(hipMalloc((void **)&flag, sizeof(uint32_t)));
      MGX_HIP(hipMemsetAsync(table, 0, number_size(op->d.number) * npts, s));
      MGX_HIP(hipMemsetAsync(flag, 0, sizeof(uint32_t), s));
      if (op->d.number == MGX_F64)
        macro_diag_table_f64(s, op->d, map, table, flag);
      else
        macro_diag_table_f32(s, op->d, map, table, flag);
      MGX_HIP(hipMemcpyAsync(&mismatch, flag, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
      MGX_HIP(hipStreamSynchronize(s));
      MGX_HIP(hipFree(flag));
      if (mismatch == 0)
        slot = table;
      else
        MGX_HIP(hipFree(table));
      if (which == 0)
        MGX_TRACE("compute_diagonal: diagonal %s per brick item", mismatch == 0 ? "uniform" : "not uniform");
    }
  return MGX_OK;
}

int mgx_get_inverse_diagonal(mgx_operator_t op, const void **dptr)
{
  MGX_REQUIRE(op && dptr, "mgx_get_inverse_diagonal: null argument");
  MGX_REQUIRE(op->has_diag, "mgx_get_inverse_diagonal: call mgx_compute_diagonal first");
  *dptr = op->d.inv_diag;
  return MGX_OK;
}

/* ------------------------------------------------------------------------------------------
 * PreconditionChebyshev
 * ------------------------------------------------------------------------------------------ */
int mgx_smoother_create(mgx_operator_t op, double smoothing_range, int degree, int eig_cg_n_iterations,
                        mgx_smoother_t *out)
{
  MGX_REQUIRE(op && out, "mgx_smoother_create: null argument");
  MGX_REQUIRE(eig_cg_n_iterations > 2, "mgx_smoother_create: eig_cg_n_iterations must be > 2");
  if (!op->has_diag)
    MGX_TRY(mgx_compute_diagonal(op));
  mgx_context_t ctx = op->ctx;
  hipStream_t   s   = ctx->stream;
  const int     num = op->d.number;
  const size_t  n   = op->d.n_dofs, bytes = number_size(num) * n;
  std::unique_ptr<mgx_smoother_s, int (*)(mgx_smoother_t)> sm(new mgx_smoother_s, mgx_smoother_destroy);
  sm->op            = op;
  MGX_HIP(hipMalloc(&sm->x_old, bytes));
  MGX_HIP(hipMalloc(&sm->tmp, bytes));
  // estimate_eigenvalues: PCG(D^-1) on v_i = (i mod 11) - mean; Lanczos tridiagonal
  void *r = nullptr, *z = nullptr, *d = nullptr, *h = sm->tmp, *x = sm->x_old;
  struct Scratch
  {
    void *&a, *&b, *&c;
    ~Scratch()
    {
      (void)hipFree(a);
      (void)hipFree(b);
      (void)hipFree(c);
    }
  } scratch{r, z, d};
  MGX_HIP(hipMalloc(&r, bytes));
  MGX_HIP(hipMalloc(&z, bytes));
  MGX_HIP(hipMalloc(&d, bytes));
  {
    // deal.II: v_i = (global index of i) mod 11 minus the global mean
    double sc[2] = {op->start_sum, op->start_count};
    MGX_TRY(comm_allreduce(ctx, sc, 2));
    launch_index_mod11(s, num, r, op->global_index_dev, sc[0] / sc[1], n);
  }
  MGX_HIP(hipMemsetAsync(x, 0, bytes, s));
  std::vector<double> diag, off;
  double              res = 0, rz = 0, rz_old = 0, alpha = 0, alpha_old = 0, beta = 0;
  MGX_TRY(dot(ctx, num, r, r, n, &res, op->plan.get()));
  res    = std::sqrt(res);
  int it = 0;
  MGX_TRACE("smoother_create: n=%zu eig_its=%d res0=%g", n, eig_cg_n_iterations, res);
  while (it < eig_cg_n_iterations && res > 1e-10) // IterationNumberControl(n_its, 1e-10)
    {
      ++it;
      rz_old = rz;
      launch_jacobi_dot(s, num, z, op->d.inv_diag, r, n, ctx->partial_dev, ctx->result_dev);
      if (ctx->has_comm)
        MGX_TRY(dot(ctx, num, r, z, n, &rz, op->plan.get()));
      else
        MGX_TRY(read_result(ctx, &rz));
      if (it > 1)
        {
          beta = rz / rz_old;
          launch_xpby(s, num, d, z, beta, n);
        }
      else
        launch_copy_cast(s, d, num, z, num, n);
      alpha_old = alpha;
      MGX_TRY(mgx_vmult(op, h, d));
      double dh = 0;
      MGX_TRY(dot(ctx, num, d, h, n, &dh, op->plan.get()));
      alpha = rz / dh;
      launch_cg_update(s, num, x, r, d, h, alpha, n, ctx->partial_dev, ctx->result_dev);
      if (ctx->has_comm)
        MGX_TRY(dot(ctx, num, r, r, n, &res, op->plan.get())); // duplicated interface DoFs must count once
      else
        MGX_TRY(read_result(ctx, &res));
      res = std::sqrt(res);
      if (it == 1)
        diag.push_back(1. / alpha);
      else
        {
          off.push_back(std::sqrt(beta) / alpha_old);
          diag.push_back(1. / alpha + beta / alpha_old);
        }
    }
  MGX_HIP(hipStreamSynchronize(s));
  mgx_smoother_info &info = sm->info;
  info.cg_iterations      = it;
  if (diag.empty())
    info.lambda_min = info.lambda_max = 1.;
  else
    {
      double lo, hi;
      off.push_back(0.);
      tridiag_extreme((int)diag.size(), diag.data(), off.data(), lo, hi);
      info.lambda_min = lo;
      info.lambda_max = 1.2 * hi; // safety factor
    }
  const double a =
    smoothing_range > 1. ? info.lambda_max / smoothing_range : std::min(0.9 * info.lambda_max, info.lambda_min);
  if (degree < 0) // numbers::invalid_unsigned_int: Varga's estimate for eps = smoothing_range
    {
      const double actual_range = info.lambda_max / a;
      const double sigma        = (1. - std::sqrt(1. / actual_range)) / (1. + std::sqrt(1. / actual_range));
      const double eps          = smoothing_range;
      degree = 1 + (int)(std::log(1. / eps + std::sqrt(1. / eps / eps - 1.)) / std::log(1. / sigma));
    }
  MGX_TRACE("smoother_create: its=%d lambda=[%g,%g] degree=%d", it, info.lambda_min, info.lambda_max, degree);
  info.degree = degree;
  info.delta  = (info.lambda_max - a) * 0.5;
  info.theta  = (info.lambda_max + a) * 0.5;
  sm->range_a = a;
  *out        = sm.release();
  return MGX_OK;
}

int mgx_smoother_destroy(mgx_smoother_t sm)
{
  if (!sm)
    return MGX_OK;
  (void)hipStreamSynchronize(sm->op->ctx->stream);
  (void)hipFree(sm->x_old);
  (void)hipFree(sm->x_old2);
  (void)hipFree(sm->tmp);
  delete sm;
  return MGX_OK;
}

int mgx_smoother_get_info(mgx_smoother_t sm, mgx_smoother_info *info)
{
  MGX_REQUIRE(sm && info, "mgx_smoother_get_info: null argument");
  *info = sm->info;
  return MGX_OK;
}

int mgx_smoother_set_polynomial_type(mgx_smoother_t sm, int polynomial_type)
{
  MGX_REQUIRE(sm, "mgx_smoother_set_polynomial_type: null smoother");
  MGX_REQUIRE(polynomial_type == MGX_CHEBYSHEV_FIRST_KIND || polynomial_type == MGX_CHEBYSHEV_FOURTH_KIND,
              "mgx_smoother_set_polynomial_type: unknown polynomial type");
  sm->fourth_kind = polynomial_type == MGX_CHEBYSHEV_FOURTH_KIND;
  sm->info.delta  = sm->fourth_kind ? sm->info.lambda_max : (sm->info.lambda_max - sm->range_a) * 0.5;
  return MGX_OK;
}

// Levels without a brick schedule, ordered assembly on one rank with the constrained DoFs last: the update is the
// post-operation of the assembly kernel (two launches per iteration instead of three on levels that are launch-bound)
static bool cheb_in_assembly(mgx_operator_t op)
{
  return op->d.asm_start && !op->plan && op->constrained_last && !op->ctx->tun.no_fused_assembly;
}

static void cheb_assembly_iteration(mgx_smoother_t sm, void *x, const void *b, double f1, double f2, bool three_term)
{
  mgx_operator_t op = sm->op;
  ProfileBracket pb(op, three_term ? 2 : 3);
  const ChebPost post{sm->x_old, b, op->d.inv_diag, f1, f2, three_term};
  launch_cell_loop(op->ctx->stream, op->d, x, x, nullptr, op->d.n_dofs - op->d.n_constrained, &post);
}

// legacy path (levels without a brick schedule): matvec into tmp, then an elementwise update
static int cheb_loop(mgx_smoother_t sm, void *x, const void *b)
{
  const mgx_smoother_info &I = sm->info;
  mgx_operator_t           op = sm->op;
  hipStream_t              s  = op->ctx->stream;
  if (I.degree < 2 || std::fabs(I.delta) < 1e-40)
    return MGX_OK;
  double rhok = I.delta / I.theta;
  for (int k = 0; k < I.degree - 1; ++k)
    {
      double f1, f2;
      sm->next_factors(k, rhok, f1, f2);
      if (cheb_in_assembly(op))
        {
          cheb_assembly_iteration(sm, x, b, f1, f2, true);
          continue;
        }
      MGX_TRY(mgx_vmult(op, sm->tmp, x));
      launch_cheb_update(s, op->d.number, 2, x, sm->x_old, b, sm->tmp, op->d.inv_diag, f1, f2, op->d.n_dofs);
    }
  MGX_HIP(hipGetLastError());
  return MGX_OK;
}

// One fused Chebyshev iteration on a brick-scheduled level (the counterpart of
// LaplaceOperator::vmult(dst, src, before, after), laplace_operator.h:723-741, with
// PreconditionChebyshev's update as the after-operation):
//   out <- cur + f1 (cur - out) + f2 D^-1 (b - A cur);  mode 2 general, 3 without the f1 term,
//   4 with out == 0 on entry.  sm->tmp carries the partial sums of brick-surface DoFs.
static int cheb_fused_iteration(mgx_smoother_t sm, int mode, const void *cur, const void *old, void *out,
                                const void *b, double f1, double f2, double f0 = 0., const void *coarse = nullptr,
                                const uint32_t *coarse_blocks = nullptr, const TransferData *prolong_tr = nullptr)
{
  mgx_operator_t op = sm->op;
  hipStream_t    s  = op->ctx->stream;
  // the interface DoFs' partial sums of A cur sit in sm->tmp: complete them and apply the update there
  const bool fr = op->d.bricks.fr.available() && mode >= 2 && mode <= 6;
  // (reduced-colour schedule on one rank: the update of the constrained rows rides on the finish kernel)
  const bool folded = fr && !op->plan;
  // decomposed: interface DoFs and constrained rows are updated by the unpack launch of the exchange
  const ChebList post{mode, cur, b, op->d.inv_diag, old, out, f1, f2, f0, op->d.constrained, op->d.n_constrained};
  bool           post_done = false;
  MGX_TRY(brick_loop_with_exchange(
    op, mode, sm->tmp,
    [&](hipStream_t st, int g0, int g1) {
      launch_brick_loop(st, op->d, mode, cur, b, op->d.inv_diag, out, sm->tmp, f1, f2, old, f0, const_cast<void *>(coarse),
                        coarse_blocks, g0, g1, fr);
    },
    [&](hipStream_t st) {
      if (mode == 9) // the shared DoFs start from x + P x_coarse as well (and store it: x_old of the next iteration)
        launch_interface_prolong_cheb(st, op->d.number, *prolong_tr, op->plan->shared_dev, coarse, const_cast<void *>(cur), out, b,
                                      op->d.inv_diag, f2, sm->tmp);
      else
        launch_cheb_constrained(st, op->d.number, mode, cur, out, b, op->d.inv_diag, f1, f2, op->plan->shared_dev,
                                op->plan->n_shared, sm->tmp, old, f0);
    },
    fr,
    [&](hipStream_t st, uint32_t first, uint32_t count) {
      // one rank: one call for the whole surface list, which takes the constrained rows along
      launch_surf_finish(st, op->d, mode, first, count, sm->tmp, cur, out, b, op->d.inv_diag, old, f1, f2, f0,
                         folded ? op->d.constrained : nullptr, folded ? op->d.n_constrained : 0u);
    },
    (op->plan && mode != 9 && !op->ctx->tun.exchange_unfused) ? &post : nullptr, &post_done));
  if (!folded && !post_done)
    launch_cheb_constrained(s, op->d.number, mode, cur, out, b, op->d.inv_diag, f1, f2, op->d.constrained,
                            op->d.n_constrained, nullptr, old, f0);
  MGX_HIP(hipGetLastError());
  return MGX_OK;
}

// vmult (zero start) / step (nonzero start) on a brick-scheduled level.  Every fused iteration
// reads the current iterate (gathered by the cell loop) and x_old, and writes the new iterate into
// a buffer that is not the current one.  The targets rotate over the caller's vector X and the
// smoother's buffers Y (and Z for an odd number of iterations from a nonzero start) such that the
// LAST iterate lands in X: no pointer swap (deal.II swaps solution/solution_old) and no copy, and
// all pointers stay fixed from call to call (which lets the coarse part of the V-cycle be
// replayed as a HIP graph).
// prolong_coarse / prolong_blocks (step only): the coarse-grid correction P x_coarse is added to x on
// the fly by the first iteration (mode 9) instead of by a prolongation kernel before the call
static int smoother_apply(mgx_smoother_t sm, void *x, const void *b, bool is_step, const void *prolong_coarse = nullptr,
                          const uint32_t *prolong_blocks = nullptr, const TransferData *prolong_tr = nullptr)
{
  const mgx_smoother_info &I  = sm->info;
  mgx_operator_t           op = sm->op;
  hipStream_t              s  = op->ctx->stream;
  const int                num = op->d.number;
  const size_t             n   = op->d.n_dofs;
  // levels without a brick schedule: matvec + elementwise update
  if (!op->d.bricks.available())
    {
      if (is_step && cheb_in_assembly(op))
        cheb_assembly_iteration(sm, x, b, 0., sm->first_factor(), false);
      else if (is_step)
        {
          MGX_TRY(mgx_vmult(op, sm->tmp, x));
          launch_cheb_update(s, num, 1, x, sm->x_old, b, sm->tmp, op->d.inv_diag, 0., sm->first_factor(), n);
        }
      else
        launch_cheb_update(s, num, 0, x, sm->x_old, b, nullptr, op->d.inv_diag, 0., sm->first_factor(), n);
      return cheb_loop(sm, x, b);
    }
  const bool three_term = I.degree >= 2 && std::fabs(I.delta) >= 1e-40;
  const int  n_loop     = three_term ? I.degree - 1 : 0; // iterations of the three-term recurrence
  void      *X = x, *Y = sm->x_old;
  if (!is_step && n_loop >= 1 && op->d.separable && !op->ctx->tun.no_fused_init)
    {
      // Zero initial guess: x_1 = (1/theta) D^-1 b is not stored.  The first loop iteration
      // evaluates it while gathering (mode 5), the second one again as its x_old (mode 6); from the
      // third on both operands are stored iterates.  Targets alternate so that the last is X.
      const double f0   = sm->first_factor();
      double       rhok = I.delta / I.theta;
      void        *cur = nullptr, *old = nullptr;
      for (int k = 0; k < n_loop; ++k)
        {
          double f1, f2;
          sm->next_factors(k, rhok, f1, f2);
          void *out = ((n_loop - 1 - k) % 2 == 0) ? X : Y;
          MGX_TRY(cheb_fused_iteration(sm, k == 0 ? 5 : (k == 1 ? 6 : 2), cur, old, out, b, f1, f2, f0));
          old = cur;
          cur = out;
        }
      return MGX_OK;
    }
  if (!is_step)
    {
      // x_1 = (1/theta) D^-1 b goes where an alternation over {X,Y} ends in X
      void *cur = (n_loop % 2 == 0) ? X : Y, *old = nullptr;
      launch_cheb_init(s, num, cur, b, op->d.inv_diag, sm->first_factor(), n);
      double rhok = I.delta / I.theta;
      for (int k = 0; k < n_loop; ++k)
        {
          double f1, f2;
          sm->next_factors(k, rhok, f1, f2);
          void *out = (cur == X) ? Y : X;
          MGX_TRY(cheb_fused_iteration(sm, k == 0 ? 4 : 2, cur, old, out, b, f1, f2)); // k = 0: x_0 = 0
          old = cur;
          cur = out;
        }
      return MGX_OK;
    }
  // step(): 1 + n_loop iterations starting from X
  const int T = 1 + n_loop;
  if (T % 2 == 1 && T >= 3 && !sm->x_old2)
    MGX_HIP(hipMalloc(&sm->x_old2, number_size(num) * n));
  void  *Z = sm->x_old2, *cur = X, *old = nullptr;
  double rhok = I.delta / I.theta;
  for (int k = 1; k <= T; ++k)
    {
      void *out;
      if (T % 2 == 0)
        out = (cur == X) ? Y : X;
      else if (T == 1)
        out = Y; // copied back below
      else
        out = k == 1 ? Y : (k == 2 ? Z : (k == 3 ? X : ((cur == X) ? Y : X)));
      if (k == 1)
        MGX_TRY(cheb_fused_iteration(sm, prolong_blocks ? 9 : 3, cur, nullptr, out, b, 0., sm->first_factor(), 0.,
                                     prolong_coarse, prolong_blocks, prolong_tr));
      else
        {
          double f1, f2;
          sm->next_factors(k - 2, rhok, f1, f2);
          MGX_TRY(cheb_fused_iteration(sm, 2, cur, old, out, b, f1, f2));
        }
      old = cur;
      cur = out;
    }
  if (cur != X)
    launch_copy_cast(s, X, num, cur, num, n);
  return MGX_OK;
}

int mgx_smoother_vmult(mgx_smoother_t sm, void *x, const void *b)
{
  MGX_REQUIRE(sm && x && b && x != b, "mgx_smoother_vmult: bad argument");
  return smoother_apply(sm, x, b, false);
}

int mgx_smoother_step(mgx_smoother_t sm, void *x, const void *b)
{
  MGX_REQUIRE(sm && x && b && x != b, "mgx_smoother_step: bad argument");
  return smoother_apply(sm, x, b, true);
}

/* ------------------------------------------------------------------------------------------
 * MGTransferMatrixFree (one level pair)
 * ------------------------------------------------------------------------------------------ */
// Fused level transfers on a decomposed mesh (TransferData::ifr_*, ifp_*): the brick loop restricts / prolongates the
// DoFs it completes itself; the DoFs on the rank interface are completed after the exchange, by list kernels that
// need their rows of P spelled out.  P is interpolatory: a fine DoF on a shared mesh entity couples to coarse DoFs of
// that entity only, so the first parent found holds its whole row.  idxc: the coarse operator's index table (host).
static int build_interface_transfer(mgx_transfer_s *tr, const mgx_transfer_desc *desc, const std::vector<uint32_t> &idxc)
{
  mgx_operator_t      fine = tr->fine, coarse = tr->coarse;
  const int           p = fine->d.p, n = p + 1;
  const uint32_t      npar = coarse->d.n_cells, nsh = fine->plan->n_shared;
  const ExchangePlan &pl = *fine->plan;
  std::vector<uint32_t> shared(nsh), pos(fine->d.n_dofs, kInvalid);
  if (nsh)
    MGX_HIP(hipMemcpy(shared.data(), pl.shared_dev, sizeof(uint32_t) * nsh, hipMemcpyDeviceToHost));
  for (uint32_t i = 0; i < nsh; ++i)
    pos[shared[i]] = i;
  std::vector<uint8_t> foreign(nsh, 0), done(nsh, 0);
  for (uint32_t d : pl.not_owned_host)
    if (pos[d] != kInvalid)
      foreign[pos[d]] = 1;
  std::vector<uint32_t> idxf(27 * (size_t)fine->d.n_cells);
  MGX_HIP(hipMemcpy(idxf.data(), fine->d.idx27, sizeof(uint32_t) * idxf.size(), hipMemcpyDeviceToHost));
  auto dof_of = [p](const uint32_t *ind, int ix, int iy, int iz) {
    const int      i[3] = {ix, iy, iz};
    int            code[3], off[3], len[3];
    for (int d = 0; d < 3; ++d)
      {
        code[d] = i[d] == 0 ? 0 : (i[d] == p ? 2 : 1);
        off[d]  = code[d] == 1 ? i[d] - 1 : 0;
        len[d]  = code[d] == 1 ? p - 1 : 1;
      }
    const uint32_t base = ind[(code[2] * 3 + code[1]) * 3 + code[0]];
    return base == kInvalid ? kInvalid : base + (uint32_t)((off[2] * len[1] + off[1]) * len[0] + off[0]);
  };
  struct Entry
  {
    uint32_t si, cdof;
    double   w;
  };
  std::vector<Entry> ent;
  for (uint32_t pc = 0; pc < npar; ++pc)
    for (int ch = 0; ch < 8; ++ch)
      {
        const uint32_t *indf = &idxf[27 * (size_t)desc->children[8 * (size_t)pc + ch]];
        bool            any = false;
        for (int e = 0; e < 27 && !any; ++e)
          any = indf[e] != kInvalid && indf[e] < fine->d.n_dofs && pos[indf[e]] != kInvalid &&
                (p > 1 || (e % 3 != 1 && (e / 3) % 3 != 1 && e / 9 != 1));
        if (!any)
          continue;
        for (int iz = 0; iz < n; ++iz)
          for (int iy = 0; iy < n; ++iy)
            for (int ix = 0; ix < n; ++ix)
              {
                const uint32_t d = dof_of(indf, ix, iy, iz);
                if (d == kInvalid || pos[d] == kInvalid || done[pos[d]])
                  continue;
                const uint32_t si = pos[d];
                done[si]          = 1;
                const int F[3]    = {(ch & 1) * p + ix, ((ch >> 1) & 1) * p + iy, ((ch >> 2) & 1) * p + iz};
                for (int az = 0; az < n; ++az)
                  for (int ay = 0; ay < n; ++ay)
                    for (int ax = 0; ax < n; ++ax)
                      {
                        const double w = desc->prolong_1d[F[0] * n + ax] * desc->prolong_1d[F[1] * n + ay] *
                                         desc->prolong_1d[F[2] * n + az];
                        if (std::fabs(w) < 1e-14)
                          continue;
                        const uint32_t c = dof_of(&idxc[27 * (size_t)pc], ax, ay, az);
                        if (c != kInvalid)
                          ent.push_back({si, c, w});
                      }
              }
      }
  for (uint32_t i = 0; i < nsh; ++i)
    if (!done[i])
      return fail(MGX_ERR_INVALID_ARGUMENT, "mgx_transfer_create: a shared DoF of the fine level lies in no cell");
  const int    num = fine->d.number;
  const size_t wsz = number_size(num);
  auto upload_w = [&](const std::vector<double> &w, void **dev) -> int {
    MGX_HIP(hipMalloc(dev, wsz * std::max<size_t>(1, w.size())));
    if (num == MGX_F64)
      MGX_HIP(hipMemcpy(*dev, w.data(), 8 * w.size(), hipMemcpyHostToDevice));
    else
      {
        std::vector<float> wf(w.begin(), w.end());
        MGX_HIP(hipMemcpy(*dev, wf.data(), 4 * wf.size(), hipMemcpyHostToDevice));
      }
    return MGX_OK;
  };
  auto upload_u = [&](const std::vector<uint32_t> &v, uint32_t **dev) -> int {
    MGX_HIP(hipMalloc((void **)dev, sizeof(uint32_t) * std::max<size_t>(1, v.size())));
    MGX_HIP(hipMemcpy(*dev, v.data(), sizeof(uint32_t) * v.size(), hipMemcpyHostToDevice));
    return MGX_OK;
  };
  // prolongation: rows in the order of the shared list
  {
    std::stable_sort(ent.begin(), ent.end(), [](const Entry &a, const Entry &b) { return a.si < b.si; });
    std::vector<uint32_t> start(nsh + 1, 0), cd(ent.size());
    std::vector<double>   w(ent.size());
    for (const Entry &e : ent)
      start[e.si + 1]++;
    for (uint32_t i = 0; i < nsh; ++i)
      start[i + 1] += start[i];
    for (size_t k = 0; k < ent.size(); ++k)
      {
        cd[k] = ent[k].cdof;
        w[k]  = ent[k].w;
      }
    MGX_TRY(upload_u(start, &tr->d.ifp_start));
    MGX_TRY(upload_u(cd, &tr->d.ifp_cdof));
    MGX_TRY(upload_w(w, &tr->d.ifp_w));
    tr->d.n_ifp = nsh;
  }
  // restriction: rows by coarse DoF, only the fine DoFs this rank owns (the coarse sums are added over the ranks)
  {
    std::vector<Entry> own;
    for (const Entry &e : ent)
      if (!foreign[e.si])
        own.push_back(e);
    std::stable_sort(own.begin(), own.end(), [](const Entry &a, const Entry &b) { return a.cdof < b.cdof; });
    std::vector<uint32_t> cdof, start, fd(own.size());
    std::vector<double>   w(own.size());
    for (size_t k = 0; k < own.size(); ++k)
      {
        if (k == 0 || own[k].cdof != own[k - 1].cdof)
          {
            cdof.push_back(own[k].cdof);
            start.push_back((uint32_t)k);
          }
        fd[k] = shared[own[k].si];
        w[k]  = own[k].w;
      }
    start.push_back((uint32_t)own.size());
    MGX_TRY(upload_u(cdof, &tr->d.ifr_cdof));
    MGX_TRY(upload_u(start, &tr->d.ifr_start));
    MGX_TRY(upload_u(fd, &tr->d.ifr_fdof));
    MGX_TRY(upload_w(w, &tr->d.ifr_w));
    tr->d.n_ifr = (uint32_t)cdof.size();
  }
  MGX_TRACE("transfer_create: interface rows: %u shared fine DoFs, %zu entries, %u coarse rows", nsh, ent.size(), tr->d.n_ifr);
  return MGX_OK;
}

int mgx_transfer_create(mgx_operator_t coarse, mgx_operator_t fine, const mgx_transfer_desc *desc,
                        mgx_transfer_t *out)
{
  MGX_REQUIRE(coarse && fine && desc && out && desc->children && desc->prolong_1d,
              "mgx_transfer_create: null argument");
  MGX_REQUIRE(coarse->d.p == fine->d.p && coarse->d.number == fine->d.number,
              "mgx_transfer_create: level operators differ in degree or number type");
  MGX_REQUIRE(coarse->d.idx27_plain && fine->d.idx27_plain,
              "mgx_transfer_create: operators were created without idx27_plain");
  MGX_REQUIRE((uint64_t)coarse->d.n_cells * 8 == fine->d.n_cells,
              "mgx_transfer_create: fine level must have 8 children per coarse cell (uniform refinement)");
  const int      p = coarse->d.p, n = p + 1;
  const uint32_t npar = coarse->d.n_cells;
  for (size_t i = 0; i < 8 * (size_t)npar; ++i)
    if (desc->children[i] >= fine->d.n_cells)
      return fail(MGX_ERR_INVALID_ARGUMENT, "mgx_transfer_create: child index out of range");
  MGX_TRACE("transfer_create: parents=%u", npar);
  auto tr    = std::make_unique<mgx_transfer_s>();
  tr->coarse = coarse;
  tr->fine   = fine;
  tr->d.coarse = &coarse->d;
  tr->d.colour_min = coarse->ctx->tun.restrict_colour_min;
  tr->d.fine   = &fine->d;
  MGX_HIP(hipMalloc((void **)&tr->d.children, sizeof(uint32_t) * 8 * (size_t)npar));
  MGX_HIP(hipMemcpy(tr->d.children, desc->children, sizeof(uint32_t) * 8 * (size_t)npar, hipMemcpyHostToDevice));
  // weights 1/multiplicity (multiplicity = number of parent patches sharing a fine DoF), stored
  // compressed as 3^3 entries per parent like deal.II does on uniform meshes
  {
    std::vector<uint32_t> idxf(27 * (size_t)fine->d.n_cells);
    MGX_HIP(hipMemcpy(idxf.data(), fine->d.idx27_plain, sizeof(uint32_t) * idxf.size(), hipMemcpyDeviceToHost));
    std::vector<uint8_t> cnt(fine->d.n_dofs, 0);
    // representative fine DoF of patch entity e = (ca,cb,cc): a child vertex lying in it (low
    // patch face: child 0's low vertex; patch interior: the mid plane = child 0's high vertex;
    // high patch face: child 1's high vertex) -- vertices carry exactly one DoF for every p
    auto rep = [&](uint32_t pc, int e) {
      const int      ca = e % 3, cb = (e / 3) % 3, cc = e / 9;
      const int      ch = (ca == 2) | ((cb == 2) << 1) | ((cc == 2) << 2);
      const uint32_t fc = desc->children[8 * (size_t)pc + ch];
      return idxf[27 * (size_t)fc + 9 * (cc ? 2 : 0) + 3 * (cb ? 2 : 0) + (ca ? 2 : 0)];
    };
    for (uint32_t pc = 0; pc < npar; ++pc)
      for (int e = 0; e < 27; ++e)
        cnt[rep(pc, e)]++;
    std::vector<uint8_t> shift(27 * (size_t)npar, 0);
    // Multiplicities other than 1/2/4/8 (three blocks of a multi-block mesh around an edge): the
    // restriction then uses OWNER weights -- a shared fine DoF is restricted by the one parent that
    // owns its entity, with weight 1.  Any weights that add up to one over the parents of a fine
    // DoF give the same R = P^T (P is interpolatory: a fine DoF on a shared entity only couples to
    // coarse DoFs on that entity, which all its parents hold); only the summation order differs.
    bool owner_weights = false;
    for (uint32_t pc = 0; pc < npar; ++pc)
      for (int e = 0; e < 27; ++e)
        {
          const uint8_t c = cnt[rep(pc, e)];
          if (c != 1 && c != 2 && c != 4 && c != 8)
            owner_weights = true;
          shift[27 * (size_t)pc + e] = c == 1 ? 0 : (c == 2 ? 1 : (c == 4 ? 2 : 3));
        }
    // (on a decomposed mesh the local counts miss the parents of other ranks: without the caller's global
    // weight_shift every rank uses owner weights, which need no count at all)
    if (coarse->plan && !desc->weight_shift)
      owner_weights = true;
    if (owner_weights)
      {
        if (desc->weight_shift)
          return fail(MGX_ERR_UNSUPPORTED, "mgx_transfer_create: fine DoF multiplicities other than 1/2/4/8 together with "
                                           "weight_shift");
        std::fill(shift.begin(), shift.end(), (uint8_t)0);
        tr->d.owner_weights = true;
      }
    if (desc->weight_shift) // multiplicities that count the parents of other ranks as well
      {
        for (size_t i = 0; i < shift.size(); ++i)
          {
            if (desc->weight_shift[i] > 3 || desc->weight_shift[i] < shift[i])
              return fail(MGX_ERR_INVALID_ARGUMENT, "mgx_transfer_create: inconsistent weight_shift");
            shift[i] = desc->weight_shift[i];
          }
      }
    MGX_HIP(hipMalloc((void **)&tr->d.weight_shift, shift.size()));
    MGX_HIP(hipMemcpy(tr->d.weight_shift, shift.data(), shift.size(), hipMemcpyHostToDevice));
    // ownership of the fine entities for the atomic-free prolongation: first cell in cell order
    std::vector<uint32_t> own(fine->d.n_cells, 0u);
    {
      std::vector<uint8_t> seen(fine->d.n_dofs, 0);
      for (uint32_t c = 0; c < fine->d.n_cells; ++c)
        for (int e = 0; e < 27; ++e)
          {
            const int size = (e % 3 == 1 ? p - 1 : 1) * ((e / 3) % 3 == 1 ? p - 1 : 1) * (e / 9 == 1 ? p - 1 : 1);
            if (size == 0)
              continue;
            const uint32_t base = idxf[27 * (size_t)c + e];
            if (!seen[base])
              {
                seen[base] = 1;
                own[c] |= 1u << e;
              }
          }
    }
    MGX_HIP(hipMalloc((void **)&tr->d.own27, sizeof(uint32_t) * own.size()));
    MGX_HIP(hipMemcpy(tr->d.own27, own.data(), sizeof(uint32_t) * own.size(), hipMemcpyHostToDevice));
    // patch table of the pipelined kernels (mgx_transfer.hip): the 5^3 mesh entities of the
    // children patch of every parent.  Patch entity layer 0..4 along a direction = (child 0:
    // codes 0,1,2 ; child 1: codes 0,1,2) with child 0's code 2 and child 1's code 0 coinciding.
    if (fine->d.n_dofs < (1u << 29) && !coarse->ctx->tun.transfer_v1)
      {
        std::vector<uint32_t> patch(125 * (size_t)npar);
        bool                  consistent = true;
        // owner weights on a decomposed mesh: a fine entity that a lower rank holds as well is restricted
        // there; here it enters with weight 0 (the coarse sums are added over the ranks afterwards)
        std::vector<uint8_t> foreign;
        if (tr->d.owner_weights && fine->plan)
          {
            foreign.assign(fine->d.n_dofs, 0);
            for (uint32_t i : fine->plan->not_owned_host)
              foreign[i] = 1;
          }
#pragma omp parallel for schedule(static)
        for (uint32_t pc = 0; pc < npar; ++pc)
          for (int e = 0; e < 125; ++e)
            {
              const int el[3] = {e % 5, (e / 5) % 5, e / 25};
              int       nopt[3], bit[3][2], code[3][2], cls[3], size = 1;
              for (int d = 0; d < 3; ++d)
                {
                  nopt[d]    = el[d] == 2 ? 2 : 1;
                  bit[d][0]  = el[d] > 2;
                  code[d][0] = el[d] - 2 * bit[d][0];
                  bit[d][1]  = 1; // layer 2 seen from child 1: its code 0
                  code[d][1] = 0;
                  cls[d]     = el[d] == 0 ? 0 : (el[d] == 4 ? 2 : 1);
                  size *= (el[d] % 2 == 1) ? p - 1 : 1;
                }
              uint32_t base = 0;
              bool     owned = false, first = true;
              for (int oz = 0; oz < nopt[2]; ++oz)
                for (int oy = 0; oy < nopt[1]; ++oy)
                  for (int ox = 0; ox < nopt[0]; ++ox)
                    {
                      const int      ch = bit[0][ox] | (bit[1][oy] << 1) | (bit[2][oz] << 2);
                      const int      ce = (code[2][oz] * 3 + code[1][oy]) * 3 + code[0][ox];
                      const uint32_t fc = desc->children[8 * (size_t)pc + ch];
                      const uint32_t b  = idxf[27 * (size_t)fc + ce];
                      if (first)
                        base = b;
                      else if (b != base)
                        consistent = false; // children do not share the entities of the parent's mid planes
                      first = false;
                      owned = owned || ((own[fc] >> ce) & 1u);
                    }
              uint32_t sh = shift[27 * (size_t)pc + (cls[2] * 3 + cls[1]) * 3 + cls[0]];
              if (tr->d.owner_weights) // the field then says who restricts the entity: 0 = this parent, 1 = another one
                sh = (owned && size != 0 && (foreign.empty() || !foreign[base])) ? 0u : 1u;
              patch[125 * (size_t)pc + e] =
                size == 0 ? 0u : (base | (sh << 29) | ((owned ? 1u : 0u) << 31));
            }
        if (consistent)
          {
            MGX_HIP(hipMalloc((void **)&tr->d.patch, sizeof(uint32_t) * patch.size()));
            MGX_HIP(hipMemcpy(tr->d.patch, patch.data(), sizeof(uint32_t) * patch.size(), hipMemcpyHostToDevice));
            int cus = 256;
            (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, coarse->ctx->device);
            tr->d.n_cus = (uint32_t)std::max(1, cus);
            // colouring of the coarse cells by index mod 8 (the parity colouring of a Morton-ordered
            // mesh): valid if no two cells of one colour share a mesh entity
            if (npar % 8 == 0 && !coarse->ctx->tun.restrict_atomic)
              {
                std::vector<uint32_t> idxc(27 * (size_t)npar);
                MGX_HIP(hipMemcpy(idxc.data(), coarse->d.idx27_plain, sizeof(uint32_t) * idxc.size(), hipMemcpyDeviceToHost));
                std::vector<uint8_t> mask(coarse->d.n_dofs, 0);
                bool                 ok = true;
                for (uint32_t c = 0; c < npar && ok; ++c)
                  for (int e = 0; e < 27; ++e)
                    {
                      const int size = (e % 3 == 1 ? p - 1 : 1) * ((e / 3) % 3 == 1 ? p - 1 : 1) * (e / 9 == 1 ? p - 1 : 1);
                      if (size == 0 || e == 13)
                        continue;
                      uint8_t      &m   = mask[idxc[27 * (size_t)c + e]];
                      const uint8_t bit = (uint8_t)(1u << (c & 7u));
                      if (m & bit)
                        ok = false;
                      m |= bit;
                    }
                tr->d.coarse_coloured = ok;
              }
          }
        else
          MGX_TRACE("transfer_create: children patches inconsistent, first-version kernels used");
      }
  }
  // 1D prolongation matrix into the basis blocks of both operators (the transfer kernels read the
  // coarse one, the fused residual + restriction of the fine level's cell loop the fine one)
  const size_t np1 = (size_t)(2 * p + 1) * n;
  // ... and its even-odd form for the line products of the fused forms (Basis1D::P1eo): the embedding of a parent
  // into its two children is symmetric under reversal of both indices for every node set that is
  std::vector<double> p1eo;
  bool                p1_symmetric = true;
  {
    const double *P1 = desc->prolong_1d;
    const int     nh = (p + 1) / 2;
    double        scale = 0;
    for (size_t i = 0; i < np1; ++i)
      scale = std::max(scale, std::fabs(P1[i]));
    for (int a = 0; a <= 2 * p; ++a)
      for (int j = 0; j <= p; ++j)
        if (std::fabs(P1[a * n + j] - P1[(2 * p - a) * n + (p - j)]) > 1e-12 * scale)
          p1_symmetric = false;
    p1eo.assign((size_t)(2 * p + 1) * nh + (p + 1), 0.);
    for (int a = 0; a <= p; ++a)
      for (int j = 0; j < nh; ++j)
        {
          p1eo[(size_t)a * nh + j] = 0.5 * (P1[a * n + j] + P1[a * n + p - j]);
          if (a < p)
            p1eo[(size_t)(p + 1 + a) * nh + j] = 0.5 * (P1[a * n + j] - P1[a * n + p - j]);
        }
    if (p % 2 == 0)
      for (int a = 0; a <= p; ++a)
        p1eo[(size_t)(2 * p + 1) * nh + a] = P1[a * n + p / 2];
    if (!p1_symmetric)
      MGX_TRACE("transfer_create: 1D embedding not symmetric under reversal, no fused transfer forms");
  }
  for (mgx_operator_t o : {coarse, fine})
    {
      if (o->d.number == MGX_F64)
        {
          MGX_HIP(hipMemcpy((char *)o->d.basis + offsetof(Basis1D<double>, P1), desc->prolong_1d,
                            sizeof(double) * np1, hipMemcpyHostToDevice));
          MGX_HIP(hipMemcpy((char *)o->d.basis + offsetof(Basis1D<double>, P1eo), p1eo.data(), sizeof(double) * p1eo.size(),
                            hipMemcpyHostToDevice));
        }
      else
        {
          std::vector<float> pf(desc->prolong_1d, desc->prolong_1d + np1);
          MGX_HIP(hipMemcpy((char *)o->d.basis + offsetof(Basis1D<float>, P1), pf.data(), sizeof(float) * np1,
                            hipMemcpyHostToDevice));
          std::vector<float> pe(p1eo.begin(), p1eo.end());
          MGX_HIP(hipMemcpy((char *)o->d.basis + offsetof(Basis1D<float>, P1eo), pe.data(), sizeof(float) * pe.size(),
                            hipMemcpyHostToDevice));
        }
    }
  // Fused residual + restriction (mgx_brick.hip, mode 7): needs the fine level on the brick
  // schedule in its separable form, children in forest order (cell c is child c % 8 of parent
  // c / 8, so that a brick's cells are the children of PB^3 sibling parents) and a single rank.
  // (Degree 7 ran the separate kernels until the line products of the embedding took their even-odd form in round 4:
  // V-cycle at 64^3 cells 10.69 ms separate, 9.70 ms fused; before: 8.86 / 9.36 ms at the clocks of round 2.  Degree 8
  // was in the same position while its fused forms spilled.  The option "force_fused_transfers" is a no-op now.)
  const bool fused_pays = p1_symmetric;
  // (decomposed mesh: both levels decomposed alike -- not across an agglomeration -- and interface rows, above)
  if (fine->d.bricks.available() && fine->d.separable && (!fine->plan == !coarse->plan) && fused_pays &&
      !coarse->ctx->tun.no_fused_restrict && !(fine->plan && coarse->ctx->tun.no_fused_decomposed))
    {
      bool forest = true;
      for (size_t i = 0; i < 8 * (size_t)npar && forest; ++i)
        forest = desc->children[i] == (uint32_t)i;
      const int      PB = p <= 4 ? 2 : 1, CE1 = 2 * PB + 1, CE3 = CE1 * CE1 * CE1;
      const uint32_t nb = fine->d.bricks.n_bricks, ppb = PB * PB * PB; // parents per brick
      if (forest && (uint64_t)nb * ppb == npar && fine->d.bricks.order.size() == nb)
        {
          std::vector<uint32_t> idxc(27 * (size_t)npar);
          MGX_HIP(hipMemcpy(idxc.data(), coarse->d.idx27, sizeof(uint32_t) * idxc.size(), hipMemcpyDeviceToHost));
          std::vector<uint32_t> idxp(27 * (size_t)npar);
          MGX_HIP(hipMemcpy(idxp.data(), coarse->d.idx27_plain, sizeof(uint32_t) * idxp.size(), hipMemcpyDeviceToHost));
          std::vector<uint32_t> tab((size_t)nb * CE3);
          bool                  consistent = true;
          for (uint32_t sb = 0; sb < nb; ++sb)
            {
              const uint32_t b = fine->d.bricks.order[sb]; // brick in cell order: parents ppb*b ...
              for (int e = 0; e < CE3; ++e)
                {
                  const int el[3] = {e % CE1, (e / CE1) % CE1, e / (CE1 * CE1)};
                  int       nopt[3], bit[3][2], code[3][2];
                  for (int d = 0; d < 3; ++d)
                    {
                      // entity layer el of PB parents in a row: parent el/2 (last layer: the previous
                      // parent's high side); the layer between two parents is seen from both
                      nopt[d]    = (PB == 2 && el[d] == 2) ? 2 : 1;
                      bit[d][0]  = (PB == 2 && el[d] > 2) ? 1 : 0;
                      code[d][0] = el[d] - 2 * bit[d][0];
                      bit[d][1]  = 1;
                      code[d][1] = 0;
                    }
                  uint32_t word = 0, key = 0;
                  bool     first = true;
                  for (int oz = 0; oz < nopt[2]; ++oz)
                    for (int oy = 0; oy < nopt[1]; ++oy)
                      for (int ox = 0; ox < nopt[0]; ++ox)
                        {
                          const int      q  = bit[0][ox] | (bit[1][oy] << 1) | (bit[2][oz] << 2);
                          const int      ce = (code[2][oz] * 3 + code[1][oy]) * 3 + code[0][ox];
                          const uint32_t pc = ppb * b + (uint32_t)q;
                          if (first)
                            {
                              word = idxc[27 * (size_t)pc + ce];
                              key  = idxp[27 * (size_t)pc + ce];
                            }
                          else if (idxp[27 * (size_t)pc + ce] != key)
                            consistent = false;
                          first = false;
                        }
                  tab[(size_t)sb * CE3 + e] = word;
                }
            }
          if (consistent && fine->plan)
            {
              const int status = build_interface_transfer(tr.get(), desc, idxc);
              if (status != MGX_OK)
                {
                  (void)mgx_transfer_destroy(tr.release());
                  return status;
                }
            }
          if (consistent)
            {
              MGX_HIP(hipMalloc((void **)&tr->d.coarse_blocks, sizeof(uint32_t) * tab.size()));
              MGX_HIP(hipMemcpy(tr->d.coarse_blocks, tab.data(), sizeof(uint32_t) * tab.size(), hipMemcpyHostToDevice));
            }
          // scratch form of the fused residual + restriction: every brick stores its (PB p + 1)^3 restricted values
          // in a block of its own; the coarse vector is assembled from the blocks in brick order
          const uint64_t CN = (uint64_t)PB * p + 1, NC = CN * CN * CN;
          // (levels on the reduced-colour schedules only: there a colour launch is as long as a brick's latency chain and
          // one launch instead of eight pays -- 17 M DoFs 179 -> 164 us; on the finest level of C2 the blocks' extra traffic
          // costs more than the seven launch ramps it saves, 1.184 against 1.138 ms)
          if (consistent && (uint64_t)nb * NC < 0xFFFFFFF0ull && !coarse->ctx->tun.no_restrict_scratch &&
              nb <= coarse->ctx->tun.free_max_bricks)
            {
              auto layer = [p](int a, int &e, int &o, int &len) {
                const int q = a / p, rr = a - q * p;
                e           = 2 * q + (rr != 0);
                o           = rr ? rr - 1 : 0;
                len         = rr ? p - 1 : 1;
              };
              const uint32_t        ncd = coarse->d.n_dofs;
              std::vector<uint32_t> start(ncd + 1, 0), dof((size_t)nb * NC, kInvalid);
#pragma omp parallel for schedule(static)
              for (uint32_t sb = 0; sb < nb; ++sb)
                for (uint32_t l = 0; l < NC; ++l)
                  {
                    const int x = (int)(l % CN), y = (int)((l / CN) % CN), z = (int)(l / (CN * CN));
                    int       ex, ey, ez, ox, oy, oz, nx, ny, nz;
                    layer(x, ex, ox, nx);
                    layer(y, ey, oy, ny);
                    layer(z, ez, oz, nz);
                    const uint32_t w = tab[(size_t)sb * CE3 + (size_t)((ez * CE1 + ey) * CE1 + ex)];
                    if (w != kInvalid)
                      dof[(size_t)sb * NC + l] = w + (uint32_t)((oz * ny + oy) * nx + ox);
                  }
              for (uint32_t d : dof)
                if (d != kInvalid)
                  start[d + 1]++;
              for (uint32_t d = 0; d < ncd; ++d)
                start[d + 1] += start[d];
              std::vector<uint32_t> pos(start[ncd]), fill(start.begin(), start.end() - 1);
              for (size_t k = 0; k < dof.size(); ++k) // ascending position = ascending brick: the order of the sums
                if (dof[k] != kInvalid)
                  pos[fill[dof[k]]++] = (uint32_t)k;
              MGX_HIP(hipMalloc(&tr->d.coarse_scratch, number_size(fine->d.number) * (size_t)nb * NC));
              MGX_HIP(hipMemset(tr->d.coarse_scratch, 0, number_size(fine->d.number) * (size_t)nb * NC));
              MGX_HIP(hipMalloc((void **)&tr->d.cs_start, sizeof(uint32_t) * start.size()));
              MGX_HIP(hipMemcpy(tr->d.cs_start, start.data(), sizeof(uint32_t) * start.size(), hipMemcpyHostToDevice));
              MGX_HIP(hipMalloc((void **)&tr->d.cs_pos, sizeof(uint32_t) * std::max<size_t>(1, pos.size())));
              MGX_HIP(hipMemcpy(tr->d.cs_pos, pos.data(), sizeof(uint32_t) * pos.size(), hipMemcpyHostToDevice));
            }
        }
    }
  if (tr->d.owner_weights && !tr->d.patch)
    {
      std::unique_ptr<mgx_transfer_s, int (*)(mgx_transfer_t)> guard(tr.release(), mgx_transfer_destroy);
      return fail(MGX_ERR_UNSUPPORTED, "mgx_transfer_create: owner weights need the pipelined transfer kernels (fine level below 2^29 DoFs)");
    }
  *out = tr.release();
  return MGX_OK;
}

int mgx_transfer_destroy(mgx_transfer_t tr)
{
  if (!tr)
    return MGX_OK;
  (void)hipStreamSynchronize(tr->coarse->ctx->stream);
  (void)hipFree(tr->d.children);
  (void)hipFree(tr->d.weight_shift);
  (void)hipFree(tr->d.own27);
  (void)hipFree(tr->d.patch);
  (void)hipFree(tr->d.coarse_blocks);
  (void)hipFree(tr->d.coarse_scratch);
  (void)hipFree(tr->d.cs_start);
  (void)hipFree(tr->d.cs_pos);
  for (void *q : {(void *)tr->d.ifr_cdof, (void *)tr->d.ifr_start, (void *)tr->d.ifr_fdof, tr->d.ifr_w, (void *)tr->d.ifp_start,
                  (void *)tr->d.ifp_cdof, tr->d.ifp_w})
    (void)hipFree(q);
  (void)hipFree(tr->scratch);
  delete tr;
  return MGX_OK;
}

int mgx_prolongate(mgx_transfer_t tr, void *fine, const void *coarse, int add, int with_constraints)
{
  MGX_REQUIRE(tr && fine && coarse, "mgx_prolongate: null argument");
  launch_prolongate(tr->coarse->ctx->stream, tr->d, fine, coarse, add != 0, with_constraints != 0);
  MGX_HIP(hipGetLastError());
  return MGX_OK;
}

int mgx_restrict_and_add(mgx_transfer_t tr, void *coarse, const void *fine, int with_constraints)
{
  MGX_REQUIRE(tr && fine && coarse, "mgx_restrict_and_add: null argument");
  mgx_operator_t cop = tr->coarse;
  hipStream_t    s   = cop->ctx->stream;
  if (!cop->plan)
    {
      launch_restrict_add(s, tr->d, coarse, fine, with_constraints != 0);
      MGX_HIP(hipGetLastError());
      return MGX_OK;
    }
  // decomposed mesh: restrict into a zeroed scratch vector, sum the interface entries over the
  // ranks, then add (adding into `coarse` first would count its existing interface values once
  // per sharing rank)
  const size_t bytes = number_size(cop->d.number) * cop->d.n_dofs;
  if (!tr->scratch)
    MGX_HIP(hipMalloc(&tr->scratch, bytes));
  MGX_HIP(hipMemsetAsync(tr->scratch, 0, bytes, s));
  launch_restrict_add(s, tr->d, tr->scratch, fine, with_constraints != 0);
  MGX_TRY(exchange_add(cop, tr->scratch));
  launch_add_cast(s, coarse, cop->d.number, tr->scratch, cop->d.number, cop->d.n_dofs);
  MGX_HIP(hipGetLastError());
  return MGX_OK;
}

/* ------------------------------------------------------------------------------------------
 * MultigridSolver
 * ------------------------------------------------------------------------------------------ */
int mgx_solver_set_polynomial_type(mgx_solver_t S, int polynomial_type)
{
  MGX_REQUIRE(S, "mgx_solver_set_polynomial_type: null solver");
  // the coarsest level keeps the first kind with the degree from the tolerance (multigrid_solver.h:955-959)
  for (int l = 1; l < S->n_levels; ++l)
    MGX_TRY(mgx_smoother_set_polynomial_type(S->smooth[l], polynomial_type));
  if (S->graph_exec) // the factors are baked into the captured launches
    {
      MGX_HIP(hipStreamSynchronize(S->ctx->stream));
      (void)hipGraphExecDestroy(S->graph_exec);
      (void)hipGraphDestroy(S->graph);
      S->graph_exec  = nullptr;
      S->graph       = nullptr;
      S->graph_calls = 0;
    }
  if (S->agg_solver)
    MGX_TRY(mgx_solver_set_polynomial_type(S->agg_solver, polynomial_type));
  return MGX_OK;
}

int mgx_solver_destroy(mgx_solver_t S)
{
  if (!S)
    return MGX_OK;
  (void)hipStreamSynchronize(S->ctx->stream);
  for (auto sm : S->smooth)
    mgx_smoother_destroy(sm);
  for (auto p : S->solution)
    (void)hipFree(p);
  for (auto p : S->rhs)
    (void)hipFree(p);
  for (auto p : S->residual)
    (void)hipFree(p);
  for (auto p : S->defect)
    (void)hipFree(p);
  for (auto p : S->t)
    (void)hipFree(p);
  for (auto p : S->solution_update)
    (void)hipFree(p);
  for (auto p : S->bc_index_dev)
    (void)hipFree(p);
  for (auto p : S->bc_value_dev)
    (void)hipFree(p);
  for (auto p : S->bc_zero_dev)
    (void)hipFree(p);
  if (S->graph_exec)
    (void)hipGraphExecDestroy(S->graph_exec);
  if (S->graph)
    (void)hipGraphDestroy(S->graph);
  (void)hipFree(S->agg_map);
  (void)hipFree(S->agg_owned);
  if (S->agg_in)
    (void)hipEventDestroy(S->agg_in);
  if (S->agg_out)
    (void)hipEventDestroy(S->agg_out);
  (void)hipFree(S->cg_r);
  (void)hipFree(S->cg_z);
  (void)hipFree(S->cg_d);
  (void)hipFree(S->cg_h);
  delete S;
  return MGX_OK;
}

int mgx_solver_create(mgx_context_t ctx, const mgx_solver_desc *desc, mgx_solver_t *out)
{
  MGX_REQUIRE(ctx && desc && out, "mgx_solver_create: null argument");
  MGX_REQUIRE(desc->n_levels >= 1 && desc->matrix && desc->matrix_dp && desc->bc_count,
              "mgx_solver_create: incomplete descriptor");
  MGX_REQUIRE(desc->n_levels == 1 || (desc->transfer && desc->transfer_dp), "mgx_solver_create: missing transfers");
  MGX_REQUIRE(desc->degree_pre >= 1 && desc->n_cycles >= 1, "mgx_solver_create: bad smoother degree / cycle count");
  MGX_TRACE("solver_create: n_levels=%d", desc->n_levels);
  auto S      = std::unique_ptr<mgx_solver_s, int (*)(mgx_solver_t)>(new mgx_solver_s, mgx_solver_destroy);
  S->ctx      = ctx;
  S->n_levels = desc->n_levels;
  S->degree   = desc->degree_pre;
  S->n_cycles = desc->n_cycles;
  S->vnumber  = desc->matrix[0]->d.number;
  S->timings.assign(6 * (size_t)desc->n_levels, 0.);
  const int nl = desc->n_levels;
  S->transfer.assign(nl, nullptr);
  S->transfer_dp.assign(nl, nullptr);
  for (int l = 0; l < nl; ++l)
    {
      MGX_REQUIRE(desc->matrix[l] && desc->matrix_dp[l], "mgx_solver_create: null level operator");
      MGX_REQUIRE(desc->matrix_dp[l]->d.number == MGX_F64, "mgx_solver_create: matrix_dp must be fp64");
      MGX_REQUIRE(desc->matrix[l]->d.number == S->vnumber, "mgx_solver_create: mixed V-cycle number types");
      MGX_REQUIRE(desc->matrix[l]->d.n_dofs == desc->matrix_dp[l]->d.n_dofs, "mgx_solver_create: level size mismatch");
      S->matrix.push_back(desc->matrix[l]);
      S->matrix_dp.push_back(desc->matrix_dp[l]);
      if (l > 0)
        {
          MGX_REQUIRE(desc->transfer[l] && desc->transfer_dp[l], "mgx_solver_create: null transfer");
          S->transfer[l]    = desc->transfer[l];
          S->transfer_dp[l] = desc->transfer_dp[l];
        }
      const size_t n = desc->matrix[l]->d.n_dofs;
      double      *p = nullptr;
      void        *q = nullptr;
      MGX_HIP(hipMalloc((void **)&p, 8 * n));
      MGX_HIP(hipMemsetAsync(p, 0, 8 * n, ctx->stream));
      S->solution.push_back(p);
      MGX_HIP(hipMalloc((void **)&p, 8 * n));
      if (desc->rhs && desc->rhs[l])
        {
          MGX_HIP(hipMemcpy(p, desc->rhs[l], 8 * n, hipMemcpyHostToDevice));
          // a rank assembles the rhs over its own cells: complete the interface entries
          // (dst.compress(add) in compute_residual, laplace_operator.h:843)
          MGX_TRY(exchange_add(desc->matrix_dp[l], p));
        }
      else // assembled on the device afterwards: mgx_solver_compute_rhs
        MGX_HIP(hipMemsetAsync(p, 0, 8 * n, ctx->stream));
      S->rhs.push_back(p);
      MGX_HIP(hipMalloc((void **)&p, 8 * n));
      MGX_HIP(hipMemsetAsync(p, 0, 8 * n, ctx->stream));
      S->residual.push_back(p);
      const size_t vb = number_size(S->vnumber) * n;
      MGX_HIP(hipMalloc(&q, vb));
      MGX_HIP(hipMemsetAsync(q, 0, vb, ctx->stream));
      S->defect.push_back(q);
      MGX_HIP(hipMalloc(&q, vb));
      MGX_HIP(hipMemsetAsync(q, 0, vb, ctx->stream));
      S->t.push_back(q);
      MGX_HIP(hipMalloc(&q, vb));
      MGX_HIP(hipMemsetAsync(q, 0, vb, ctx->stream));
      S->solution_update.push_back(q);
      // inhomogeneous boundary values (multigrid_solver.h:225-253)
      const uint32_t nb = desc->bc_count[l];
      S->bc_count.push_back(nb);
      uint32_t *bi = nullptr;
      double   *bv = nullptr, *bz = nullptr;
      MGX_HIP(hipMalloc((void **)&bi, sizeof(uint32_t) * (nb + 1)));
      MGX_HIP(hipMalloc((void **)&bv, sizeof(double) * (nb + 1)));
      MGX_HIP(hipMalloc((void **)&bz, sizeof(double) * (nb + 1)));
      MGX_HIP(hipMemsetAsync(bz, 0, sizeof(double) * (nb + 1), ctx->stream));
      if (nb)
        {
          MGX_REQUIRE(desc->bc_index && desc->bc_value && desc->bc_index[l] && desc->bc_value[l],
                      "mgx_solver_create: missing boundary lists");
          for (uint32_t i = 0; i < nb; ++i)
            if (desc->bc_index[l][i] >= n)
              return fail(MGX_ERR_INVALID_ARGUMENT, "mgx_solver_create: boundary index out of range");
          MGX_HIP(hipMemcpy(bi, desc->bc_index[l], sizeof(uint32_t) * nb, hipMemcpyHostToDevice));
          MGX_HIP(hipMemcpy(bv, desc->bc_value[l], sizeof(double) * nb, hipMemcpyHostToDevice));
        }
      S->bc_index_dev.push_back(bi);
      S->bc_value_dev.push_back(bv);
      S->bc_zero_dev.push_back(bz);
    }
  // smoothers (multigrid_solver.h:269-289)
  for (int l = 0; l < nl; ++l)
    {
      mgx_smoother_t sm = nullptr;
      MGX_TRY(mgx_compute_diagonal(S->matrix[l])); // :286
      if (l > 0)
        MGX_TRY(mgx_smoother_create(S->matrix[l], 20., S->degree, 15, &sm)); // :274-278
      else
        MGX_TRY(mgx_smoother_create(S->matrix[l], 1e-3, -1, (int)std::max<uint32_t>(3u, S->matrix[l]->d.n_dofs),
                                    &sm)); // :282-284
      S->smooth.push_back(sm);
    }
  if (!ctx->tun.no_graph)
    {
      const uint32_t graph_max = ctx->tun.graph_max_dofs;
      for (int l = 0; l < nl; ++l)
        if (S->matrix[l]->d.n_dofs <= graph_max)
          S->graph_level = l;
    }
  const size_t nmax = S->matrix[nl - 1]->d.n_dofs;
  MGX_HIP(hipMalloc((void **)&S->cg_r, 8 * nmax));
  MGX_HIP(hipMalloc((void **)&S->cg_z, 8 * nmax));
  MGX_HIP(hipMalloc((void **)&S->cg_d, 8 * nmax));
  MGX_HIP(hipMalloc((void **)&S->cg_h, 8 * nmax));
  *out = S.release();
  return MGX_OK;
}

static void set_bc(mgx_solver_t S, int level, double *v, bool zero)
{
  launch_scatter_values(S->ctx->stream, MGX_F64, v, S->bc_index_dev[level],
                        zero ? S->bc_zero_dev[level] : S->bc_value_dev[level], S->bc_count[level]);
}

static int v_cycle_eager(mgx_solver_t S, int level, int my_n_cycles);

static bool graph_usable(mgx_solver_t S, int level)
{
  if (S->graph_level < 0 || level != S->graph_level || S->graph_failed || S->timing || S->ctx->has_comm)
    return false;
  if (S->ctx->profile)
    for (int l = 0; l <= level; ++l)
      if (S->matrix[l]->profiled)
        return false; // HIP-event brackets must stay outside a captured region
  return true;
}

// The V-cycle on levels <= agg_level of a decomposed hierarchy, run on the undecomposed copy of
// those levels every rank holds: the defect is summed over the ranks into the copy's defect vector
// (every DoF contributed by its owner, all others add zero: exact, identical on all ranks), the
// copy's V-cycle runs on its own context (graph replay, no exchanges), and every rank reads the
// correction of its DoFs back.  Replaces one latency-bound exchange per operator application and
// level by one allreduce per V-cycle.
static int v_cycle(mgx_solver_t S, int level, int my_n_cycles);

static int agglomerated_cycle(mgx_solver_t S, int my_n_cycles)
{
  mgx_solver_t  G   = S->agg_solver;
  mgx_context_t ctx = S->ctx;
  const int     L   = S->agg_level, Lg = L + S->agg_offset; // the seam level in the numbering of either solver
  hipStream_t   s = ctx->stream, sg = G->ctx->stream;
  const int     num = S->vnumber;
  const size_t  ng = G->matrix[Lg]->d.n_dofs, nl = S->matrix[L]->d.n_dofs;
  Stopwatch     sw(S, L, 5);
  MGX_HIP(hipMemsetAsync(G->defect[Lg], 0, number_size(num) * ng, s));
  launch_scatter_map(s, num, G->defect[Lg], S->defect[L], S->agg_map, S->agg_owned, (uint32_t)nl);
  if (ctx->use_rccl)
    {
      RcclApi &R = rccl_api();
      if (R.AllReduce(G->defect[Lg], G->defect[Lg], ng, num == MGX_F64 ? ncclDouble : ncclFloat, ncclSum, ctx->nccl, s) !=
          ncclSuccess)
        return fail(MGX_ERR_HIP, "ncclAllReduce of the agglomerated defect failed");
    }
  else
    {
      S->agg_host.resize(ng);
      if (num == MGX_F64)
        MGX_HIP(hipMemcpyAsync(S->agg_host.data(), G->defect[Lg], 8 * ng, hipMemcpyDeviceToHost, s));
      else
        {
          std::vector<float> tmp(ng);
          MGX_HIP(hipMemcpyAsync(tmp.data(), G->defect[Lg], 4 * ng, hipMemcpyDeviceToHost, s));
          MGX_HIP(hipStreamSynchronize(s));
          std::copy(tmp.begin(), tmp.end(), S->agg_host.begin());
        }
      MGX_HIP(hipStreamSynchronize(s));
      // (the callback counts in int: pieces of at most 2^30 values, so that no count is ever narrowed)
      if (!ctx->comm.allreduce_sum)
        return fail(MGX_ERR_HIP, "allreduce_sum callback failed");
      for (size_t first = 0; first < (size_t)ng; first += (size_t)1 << 30)
        {
          const size_t piece = std::min<size_t>((size_t)ng - first, (size_t)1 << 30);
          if (ctx->comm.allreduce_sum(ctx->comm.user, S->agg_host.data() + first, (int)piece) != 0)
            return fail(MGX_ERR_HIP, "allreduce_sum callback failed");
        }
      if (num == MGX_F64)
        {
          // pageable staging buffer, rewritten by the next cycle: the copy must have left it before we return
          MGX_HIP(hipMemcpyAsync(G->defect[Lg], S->agg_host.data(), 8 * ng, hipMemcpyHostToDevice, s));
          MGX_HIP(hipStreamSynchronize(s));
        }
      else
        {
          std::vector<float> tmp(S->agg_host.begin(), S->agg_host.end());
          MGX_HIP(hipMemcpyAsync(G->defect[Lg], tmp.data(), 4 * ng, hipMemcpyHostToDevice, s));
          MGX_HIP(hipStreamSynchronize(s));
        }
    }
  if (sg != s)
    {
      MGX_HIP(hipEventRecord(S->agg_in, s));
      MGX_HIP(hipStreamWaitEvent(sg, S->agg_in, 0));
    }
  MGX_TRY(v_cycle(G, Lg, my_n_cycles));
  if (sg != s)
    {
      MGX_HIP(hipEventRecord(S->agg_out, sg));
      MGX_HIP(hipStreamWaitEvent(s, S->agg_out, 0));
    }
  launch_pack(s, num, S->solution_update[L], G->solution_update[Lg], S->agg_map, (uint32_t)nl);
  MGX_HIP(hipGetLastError());
  return MGX_OK;
}

int mgx_solver_set_agglomeration(mgx_solver_t S, int level, mgx_solver_t coarse, const uint32_t *local_to_global,
                                 const uint8_t *owned, uint32_t n_local)
{
  MGX_REQUIRE(S && coarse && local_to_global && owned, "mgx_solver_set_agglomeration: null argument");
  MGX_REQUIRE(S->agg_solver == nullptr, "mgx_solver_set_agglomeration: already set");
  // (a hierarchy that lacks the coarsest levels of the whole mesh may consist of its finest level alone: the seam is then
  // that level, the only one that exists on both sides)
  MGX_REQUIRE(level >= 0 && (level < S->n_levels - 1 || (level == S->n_levels - 1 && coarse->n_levels > level + 1)),
              "mgx_solver_set_agglomeration: the finest level stays decomposed");
  // (a hierarchy whose level 0 is level k of the whole mesh -- mgx_cube_level_offset -- meets a coarse solver with k more
  // levels: its finest level is the seam either way)
  MGX_REQUIRE(coarse->n_levels >= level + 1, "mgx_solver_set_agglomeration: the coarse solver must end at `level`");
  const int offset = coarse->n_levels - 1 - level;
  MGX_REQUIRE(coarse->vnumber == S->vnumber && coarse->degree == S->degree,
              "mgx_solver_set_agglomeration: number type or smoother degree differ");
  MGX_REQUIRE(coarse->ctx != S->ctx && !coarse->ctx->has_comm,
              "mgx_solver_set_agglomeration: the coarse solver lives on a context of its own without a communicator");
  MGX_REQUIRE(n_local == S->matrix[level]->d.n_dofs, "mgx_solver_set_agglomeration: map length is not the level size");
  const uint32_t ng = coarse->matrix[level + offset]->d.n_dofs;
  for (uint32_t i = 0; i < n_local; ++i)
    if (local_to_global[i] >= ng)
      return fail(MGX_ERR_INVALID_ARGUMENT, "mgx_solver_set_agglomeration: map entry out of range");
  MGX_HIP(hipMalloc((void **)&S->agg_map, sizeof(uint32_t) * ((size_t)n_local + 1)));
  MGX_HIP(hipMalloc((void **)&S->agg_owned, (size_t)n_local + 1));
  MGX_HIP(hipMemcpy(S->agg_map, local_to_global, sizeof(uint32_t) * n_local, hipMemcpyHostToDevice));
  MGX_HIP(hipMemcpy(S->agg_owned, owned, n_local, hipMemcpyHostToDevice));
  MGX_HIP(hipEventCreateWithFlags(&S->agg_in, hipEventDisableTiming));
  MGX_HIP(hipEventCreateWithFlags(&S->agg_out, hipEventDisableTiming));
  // the copy runs in line with the decomposed levels: on their stream (no event hops around its graph)
  if (!coarse->ctx->borrowed_stream && coarse->ctx->device == S->ctx->device)
    {
      MGX_HIP(hipStreamSynchronize(coarse->ctx->stream));
      MGX_HIP(hipStreamDestroy(coarse->ctx->stream));
      coarse->ctx->stream          = S->ctx->stream;
      coarse->ctx->borrowed_stream = true;
    }
  S->agg_solver = coarse;
  S->agg_level  = level;
  S->agg_offset = offset;
  return MGX_OK;
}

// MultigridSolver::v_cycle (multigrid_solver.h:641-681), with graph replay of the coarse part
static int v_cycle(mgx_solver_t S, int level, int my_n_cycles)
{
  if (S->agg_solver && level == S->agg_level)
    return agglomerated_cycle(S, my_n_cycles);
  if (my_n_cycles != 1 || !graph_usable(S, level))
    return v_cycle_eager(S, level, my_n_cycles);
  hipStream_t s = S->ctx->stream;
  if (S->graph_exec)
    {
      // (the levels inside the replayed graph cannot be cut by host ranges: one range for the whole coarse part)
      if (S->ctx->tun.roctx)
        {
          char buf[48];
          std::snprintf(buf, sizeof(buf), "v_cycle_graph_levels_0_to_%d", level);
          range_push(S->ctx, buf);
        }
      const hipError_t e = hipGraphLaunch(S->graph_exec, s);
      range_pop(S->ctx);
      MGX_HIP(e);
      return MGX_OK;
    }
  if (S->graph_calls++ == 0)
    return v_cycle_eager(S, level, 1); // first execution eager: lazy allocations happen here
  if (hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed) != hipSuccess)
    {
      S->graph_failed = true;
      return v_cycle_eager(S, level, 1);
    }
  const int  status = v_cycle_eager(S, level, 1);
  hipError_t e      = hipStreamEndCapture(s, &S->graph);
  if (status != MGX_OK || e != hipSuccess || !S->graph ||
      hipGraphInstantiate(&S->graph_exec, S->graph, nullptr, nullptr, 0) != hipSuccess)
    {
      S->graph_failed = true;
      S->graph_exec   = nullptr;
      (void)hipGetLastError();
      return v_cycle_eager(S, level, 1); // nothing was executed during the capture
    }
  MGX_TRACE("v_cycle: levels 0..%d captured into a HIP graph", level);
  MGX_HIP(hipGraphLaunch(S->graph_exec, s));
  return MGX_OK;
}

static int v_cycle_eager(mgx_solver_t S, int level, int my_n_cycles)
{
  hipStream_t s = S->ctx->stream;
  if (level == 0)
    {
      Stopwatch sw(S, 0, 0);
      S->timings[1] += 1;
      return smoother_apply(S->smooth[0], S->solution_update[0], S->defect[0], false); // :647 (MGCoarseFromSmoother :72-91)
    }
  const size_t nc = S->matrix[level - 1]->d.n_dofs;
  for (int c = 0; c < my_n_cycles; ++c)
    {
      {
        Stopwatch sw(S, level, 5);
        if (c == 0) // :656-659
          MGX_TRY(smoother_apply(S->smooth[level], S->solution_update[level], S->defect[level], false));
        else
          MGX_TRY(smoother_apply(S->smooth[level], S->solution_update[level], S->defect[level], true));
      }
      // Levels on the one-launch schedule run the residual as one launch and the transfers as kernels of
      // their own: the fused forms below need the eight colour launches, each as long as a brick's
      // whole latency chain on such a level (2 M DoFs: 63 instead of 147 us, 52 instead of 180 us)
      const bool one_launch = S->matrix[level]->d.bricks.fr.available() && S->matrix[level]->d.bricks.fr.n_classes == 1;
      // (with the scratch form of the restriction also on the one-launch schedule: it is one launch itself)
      if (S->transfer[level]->d.coarse_blocks && (!one_launch || S->transfer[level]->d.coarse_scratch) &&
          !S->ctx->tun.no_fused_residual)
        {
          // residual and restriction in one pass of the cell loop: t only carries the partial sums
          // of brick-surface DoFs between the colour launches, the residual is never stored
          Stopwatch           sw(S, level, 0);
          mgx_operator_t      A = S->matrix[level];
          const TransferData &T = S->transfer[level]->d;
          // scratch form: the bricks store their restricted values block by block (one launch for the level), the
          // coarse defect is assembled from the blocks; otherwise they add into the zeroed coarse defect colour by
          // colour.  (The form carries no partial sums: its `partial` argument names the scratch array.)
          void *scratch = (T.coarse_scratch && !A->d.cells_form) ? T.coarse_scratch : nullptr;
          if (!scratch)
            MGX_HIP(hipMemsetAsync(S->defect[level - 1], 0, number_size(S->vnumber) * nc, s)); // :667
          {
            ProfileBracket pb(A, 7);
            // (the cell-by-cell form of a cross-check build still hands partial sums over through t)
            launch_brick_loop(s, A->d, 7, S->solution_update[level], S->defect[level], nullptr, S->t[level],
                              scratch ? scratch : (A->d.cells_form ? S->t[level] : nullptr), 0., 0., nullptr, 0.,
                              S->defect[level - 1], T.coarse_blocks);
            if (scratch)
              launch_coarse_assemble(s, A->d.number, T, S->defect[level - 1]);
          }
          if (A->plan)
            {
              // decomposed: every rank's bricks restrict their own shares of A x on all their points and b on the
              // points they complete; a DoF on the rank interface is completed by no brick, its b comes from its owner
              // (list kernel); no exchange on the fine level, the coarse defect is summed over the ranks like any
              // restricted vector
              launch_interface_restrict(s, A->d.number, T, S->defect[level - 1], S->defect[level], nullptr);
              MGX_TRY(exchange_add(S->matrix[level - 1], S->defect[level - 1]));
            }
        }
      else
        {
          {
            Stopwatch sw(S, level, 0);
            MGX_TRY(mgx_vmult_residual(S->matrix[level], S->defect[level], S->solution_update[level], S->t[level])); // :663
          }
          {
            Stopwatch sw(S, level, 1);
            MGX_HIP(hipMemsetAsync(S->defect[level - 1], 0, number_size(S->vnumber) * nc, s));         // :667
            MGX_TRY(mgx_restrict_and_add(S->transfer[level], S->defect[level - 1], S->t[level], 1)); // :668
          }
        }
      MGX_TRY(v_cycle(S, level - 1, 1)); // :671
      // :674 + :678.  On one rank with the macro-element brick loop the prolongation is not a kernel of
      // its own: the first post-smoothing iteration forms x + P x_coarse while it gathers x
      mgx_operator_t Af = S->matrix[level];
      const bool     fused_prolong = S->transfer[level]->d.coarse_blocks && Af->d.bricks.item_map && !Af->d.cells_form &&
                                 Af->d.separable && !S->ctx->tun.no_fused_prolong && S->smooth[level]->info.degree >= 1 && !one_launch &&
                                 (uint64_t)Af->d.n_dofs * number_size(Af->d.number) < 0xFFFFFFF0ull &&
                                 // (a level on the two-class schedule -- decomposed or not -- runs its Chebyshev steps in two launches + finish; the fused
                                 // prolongation form needs the eight colours, each as long as a brick's latency chain there.  p <= 4
                                 // only: at p = 8 the prolongation kernel costs 0.18 ms on the 17 M-DoF level and the fused form wins,
                                 // 0.945 against 1.004 ms)
                                 (!Af->d.bricks.fr.available() || Af->d.p > 4 || Af->d.bricks.n_bricks >= S->ctx->tun.fused_prolong_min_bricks);
      if (fused_prolong)
        {
          Stopwatch sw(S, level, 5);
          MGX_TRY(smoother_apply(S->smooth[level], S->solution_update[level], S->defect[level], true,
                                 S->solution_update[level - 1], S->transfer[level]->d.coarse_blocks, &S->transfer[level]->d));
        }
      else
        {
          {
            Stopwatch sw(S, level, 2);
            MGX_TRY(mgx_prolongate(S->transfer[level], S->solution_update[level], S->solution_update[level - 1], 1,
                                   1)); // :674
          }
          {
            Stopwatch sw(S, level, 5);
            MGX_TRY(smoother_apply(S->smooth[level], S->solution_update[level], S->defect[level], true)); // :678
          }
        }
    }
  return MGX_OK;
}

int mgx_solver_compute_rhs(mgx_solver_t S, int level, const double *rhs_q)
{
  MGX_REQUIRE(S && level >= 0 && level < S->n_levels, "mgx_solver_compute_rhs: bad argument");
  // the boundary values in a vector of their own (the level's solution vector is the caller's)
  mgx_operator_t A = S->matrix_dp[level];
  double        *u = nullptr;
  MGX_HIP(hipMalloc((void **)&u, 8 * (size_t)A->d.n_dofs));
  MGX_HIP(hipMemsetAsync(u, 0, 8 * (size_t)A->d.n_dofs, S->ctx->stream));
  set_bc(S, level, u, false);
  const int status = mgx_compute_residual(A, S->rhs[level], u, rhs_q);
  (void)hipStreamSynchronize(S->ctx->stream);
  (void)hipFree(u);
  return status;
}

int mgx_solver_solve(mgx_solver_t S, int do_analyze, double *reduction_rate, double *trace)
{
  return mgx_solver_solve_hooked(S, do_analyze, reduction_rate, trace, nullptr, nullptr);
}

int mgx_solver_solve_hooked(mgx_solver_t S, int do_analyze, double *reduction_rate, double *trace, mgx_level_hook hook,
                            void *user)
{
  MGX_REQUIRE(S, "mgx_solver_solve: null solver");
  hipStream_t s    = S->ctx->stream;
  double      rate = 1.;
  int         first_level = 1;
  if (S->agg_solver && S->agg_offset > 0)
    {
      // The levels of the whole mesh below level 0 of this hierarchy exist on the undecomposed copy only (cells of level
      // 1 dealt out to the ranks, mgx_cube_create_shell_ranks): the copy -- the same problem on every rank -- runs the
      // full multigrid cycle up to the seam, every rank takes the solution of its DoFs from there and goes on above it.
      mgx_solver_t        G  = S->agg_solver;
      const int           L  = S->agg_level, Lg = L + S->agg_offset;
      std::vector<double> gt(2 * (size_t)G->n_levels, 0.);
      MGX_TRY(mgx_solver_solve_hooked(G, do_analyze, &rate, gt.data(), nullptr, nullptr));
      launch_pack(s, MGX_F64, S->solution[L], G->solution[Lg], S->agg_map, (uint32_t)S->matrix[L]->d.n_dofs);
      if (trace)
        for (int l = 0; l <= L; ++l)
          {
            trace[2 * l]     = gt[2 * (size_t)(l + S->agg_offset)];
            trace[2 * l + 1] = gt[2 * (size_t)(l + S->agg_offset) + 1];
          }
      first_level = L + 1;
    }
  else
  {
    // coarse solver invoked twice (multigrid_solver.h:397-402)
    Stopwatch    sw(S, 0, 0);
    const size_t n0 = S->matrix[0]->d.n_dofs;
    launch_copy_cast(s, S->defect[0], S->vnumber, S->rhs[0], MGX_F64, n0);
    MGX_TRY(smoother_apply(S->smooth[0], S->t[0], S->defect[0], false));
    MGX_TRY(smoother_apply(S->smooth[0], S->t[0], S->defect[0], true));
    launch_copy_cast(s, S->solution[0], MGX_F64, S->t[0], S->vnumber, n0);
    S->timings[1] += 2;
  }
  for (int level = first_level; level < S->n_levels; ++level)
    {
      const size_t n = S->matrix[level]->d.n_dofs;
      {
        Stopwatch sw(S, level, 3);
        set_bc(S, level - 1, S->solution[level - 1], false); // :408-409
      }
      {
        Stopwatch sw(S, level, 2);
        MGX_TRY(mgx_prolongate(S->transfer_dp[level], S->solution[level], S->solution[level - 1], 0, 0)); // :415
      }
      double init_residual = 1.;
      if (do_analyze && hook) // :420-424 "error start level"
        {
          MGX_HIP(hipStreamSynchronize(s));
          hook(user, level, 0);
        }
      set_bc(S, level, S->solution[level], true); // :427-428
      {
        Stopwatch sw(S, level, 0);
        MGX_TRY(mgx_vmult_residual(S->matrix_dp[level], S->rhs[level], S->solution[level], S->residual[level])); // :432
      }
      {
        Stopwatch sw(S, level, 4);
        launch_copy_cast(s, S->defect[level], S->vnumber, S->residual[level], MGX_F64, n); // :437
      }
      if (do_analyze)
        {
          MGX_TRY(mgx_operator_l2_norm(S->matrix_dp[level], S->residual[level], &init_residual)); // :444
          if (trace)
            trace[2 * level] = init_residual;
        }
      MGX_TRY(v_cycle(S, level, S->n_cycles)); // :451
      {
        Stopwatch sw(S, level, 4);
        launch_add_cast(s, S->solution[level], MGX_F64, S->solution_update[level], S->vnumber, n); // :456
      }
      if (do_analyze)
        {
          set_bc(S, level, S->solution[level], true);                                         // :462-463
          MGX_TRY(mgx_vmult(S->matrix_dp[level], S->residual[level], S->solution[level]));    // :464
          launch_sadd(s, MGX_F64, S->residual[level], -1., 1., S->rhs[level], n);             // :465
          double res_norm = 0;
          MGX_TRY(mgx_operator_l2_norm(S->matrix_dp[level], S->residual[level], &res_norm));        // :466
          rate = std::pow(res_norm / init_residual, 1. / S->n_cycles);                        // :467
          if (trace)
            trace[2 * level + 1] = res_norm;
          if (hook) // :468-472 "error end level"
            {
              MGX_HIP(hipStreamSynchronize(s));
              hook(user, level, 1);
            }
        }
    }
  MGX_HIP(hipGetLastError());
  if (reduction_rate)
    *reduction_rate = rate;
  return MGX_OK;
}

int mgx_solver_vmult(mgx_solver_t S, double *dst, const double *src)
{
  MGX_REQUIRE(S && dst && src, "mgx_solver_vmult: null argument");
  const int    lmax = S->n_levels - 1;
  const size_t n    = S->matrix[lmax]->d.n_dofs;
  hipStream_t  s    = S->ctx->stream;
  if (S->vnumber == MGX_F64 && dst != src && lmax > S->graph_level)
    {
      // same number type: the two precision-converting copies (:503, :507) are plain copies, so
      // the cycle runs on the caller's vectors instead (src is only read on the finest level,
      // every entry of dst is written by the pre-smoother)
      void *defect = S->defect[lmax], *update = S->solution_update[lmax];
      S->defect[lmax]          = const_cast<double *>(src);
      S->solution_update[lmax] = dst;
      const int status         = v_cycle(S, lmax, 1);
      S->defect[lmax]          = defect;
      S->solution_update[lmax] = update;
      return status;
    }
  launch_copy_cast(s, S->defect[lmax], S->vnumber, src, MGX_F64, n); // :503
  MGX_TRY(v_cycle(S, lmax, 1));                                       // :505
  launch_copy_cast(s, dst, MGX_F64, S->solution_update[lmax], S->vnumber, n); // :507
  return MGX_OK;
}

int mgx_solver_solve_cg(mgx_solver_t S, unsigned int *iterations, double *reduction_rate)
{
  MGX_REQUIRE(S, "mgx_solver_solve_cg: null solver");
  const int      lmax = S->n_levels - 1;
  const size_t   n    = S->matrix[lmax]->d.n_dofs;
  mgx_context_t  ctx  = S->ctx;
  hipStream_t    s    = ctx->stream;
  mgx_operator_t A    = S->matrix_dp[lmax];
  double        *x = S->solution[lmax], *r = S->cg_r, *z = S->cg_z, *d = S->cg_d, *h = S->cg_h;
  MGX_HIP(hipMemsetAsync(x, 0, 8 * n, s)); // :488
  launch_copy_cast(s, r, MGX_F64, S->rhs[lmax], MGX_F64, n);
  double res0 = 0;
  MGX_TRY(mgx_operator_l2_norm(A, r, &res0));
  double       res = res0, rz = 0, rz_old = 0;
  unsigned int it  = 0;
  S->cg_history.assign(1, res0);
  // Mixed precision on one rank: the precision casts around the V-cycle (:503, :507) are folded into the CG
  // kernels next to them (mgx_vector.hip): the residual update writes the fp32 defect, the r.z product and the
  // direction update read the fp32 result of the cycle.  The values are those the two copies would produce.
  const bool mixed = S->vnumber == MGX_F32 && !ctx->has_comm;
  float     *r32 = (float *)S->defect[lmax];
  if (mixed)
    launch_copy_cast(s, r32, MGX_F32, r, MGX_F64, n);
  // SolverCG with ReductionControl(1000, 1e-16, 1e-9) (:486)
  while (res > 1e-16 && res > 1e-9 * res0 && it < 1000)
    {
      ++it;
      rz_old = rz;
      if (mixed)
        {
          MGX_TRY(v_cycle(S, lmax, 1)); // :505
          const float *z32 = (const float *)S->solution_update[lmax];
          launch_dot_f64_f32(s, r, z32, n, ctx->partial_dev, ctx->result_dev);
          MGX_TRY(read_result(ctx, &rz));
          if (it > 1)
            launch_xpby_f64_f32(s, d, z32, rz / rz_old, n);
          else
            launch_copy_cast(s, d, MGX_F64, z32, MGX_F32, n);
        }
      else
        {
          MGX_TRY(mgx_solver_vmult(S, z, r));
          MGX_TRY(dot(ctx, MGX_F64, r, z, n, &rz, A->plan.get()));
          if (it > 1)
            launch_xpby(s, MGX_F64, d, z, rz / rz_old, n);
          else
            launch_copy_cast(s, d, MGX_F64, z, MGX_F64, n);
        }
      MGX_TRY(mgx_vmult(A, h, d));
      double dh = 0;
      MGX_TRY(dot(ctx, MGX_F64, d, h, n, &dh, A->plan.get()));
      if (mixed)
        launch_cg_update_f32copy(s, x, r, d, h, rz / dh, n, r32, ctx->partial_dev, ctx->result_dev);
      else
      launch_cg_update(s, MGX_F64, x, r, d, h, rz / dh, n, ctx->partial_dev, ctx->result_dev);
      if (ctx->has_comm)
        MGX_TRY(dot(ctx, MGX_F64, r, r, n, &res, A->plan.get()));
      else
        MGX_TRY(read_result(ctx, &res));
      res = std::sqrt(res);
      S->cg_history.push_back(res);
    }
  if (iterations)
    *iterations = it;
  if (reduction_rate)
    *reduction_rate = it > 0 ? std::pow(res / res0, 1. / it) : 1.; // :491-492
  if (it >= 1000)
    return fail(MGX_ERR_NOT_CONVERGED, "mgx_solver_solve_cg: no convergence in 1000 iterations");
  return MGX_OK;
}

/* MultigridSolver::vmult_with_residual_update (multigrid_solver.h:516-619); out3 (may be NULL)
 * additionally receives residual.residual after the update */
static int residual_update(mgx_solver_t S, double *residual, double *update, double factor, double out[2], double *rr)
{
  const int      lmax = S->n_levels - 1;
  mgx_operator_t A    = S->matrix[lmax];
  mgx_context_t  ctx  = S->ctx;
  hipStream_t    s    = ctx->stream;
  const size_t   n    = A->d.n_dofs;
  if (!A->constrained_last)
    return fail(MGX_ERR_UNSUPPORTED, "mgx_solver_vmult_with_residual_update: the constrained DoFs must be numbered last "
                                     "(local_size_without_constraints, multigrid_solver.h:525)");
  if (ctx->has_comm)
    return fail(MGX_ERR_UNSUPPORTED, "mgx_solver_vmult_with_residual_update: single rank only");
  if (!A->cg_partials)
    {
      MGX_HIP(hipMalloc((void **)&A->cg_partials, sizeof(double) * 4 * (1u << 16)));
      MGX_HIP(hipMalloc((void **)&A->cg_result, sizeof(double) * 4));
    }
  const size_t n_free = n - A->d.n_constrained;
  launch_residual_pre(s, S->vnumber, S->defect[lmax], residual, update, factor, n); // :527-534
  MGX_TRY(v_cycle(S, lmax, 1));                                                      // :538
  const uint32_t used =
    launch_residual_post(s, S->vnumber, S->solution_update[lmax], residual, update, factor, n_free, n, A->cg_partials);
  launch_reduce4(s, A->cg_partials, used, nullptr, A->cg_result);
  double h[4];
  MGX_HIP(hipMemcpyAsync(h, A->cg_result, 4 * sizeof(double), hipMemcpyDeviceToHost, s));
  MGX_HIP(hipStreamSynchronize(s));
  out[0] = h[0];
  out[1] = h[1];
  if (rr)
    *rr = h[2];
  return MGX_OK;
}

int mgx_solver_vmult_with_residual_update(mgx_solver_t S, double *residual, double *update, double factor, double out[2])
{
  MGX_REQUIRE(S && residual && update && out && residual != update, "mgx_solver_vmult_with_residual_update: bad argument");
  return residual_update(S, residual, update, factor, out, nullptr);
}

/* PCG with the matrix-vector product merged with the vector updates (mgx_vmult_with_cg_update):
 *   z = M r (q := z) ; loop: {x += alpha p ; p = beta p + q ; q = A p ; q.p} ; alpha = r.z / p.Ap ;
 *   {r -= alpha q ; r.r} ; {z = M r written into q ; r.z} ; beta = r.z_new / r.z_old
 * The residual update and the preconditioner are what vmult_with_residual_update merges; here the
 * V-cycle runs directly on r and q (no copies into and out of the level vectors in fp64), which
 * moves fewer bytes than the merged form: 3 + 2 instead of 3 + 5 passes next to the V-cycle.
 * Same iterates as mgx_solver_solve_cg up to round-off. */
int mgx_solver_solve_cg_fused(mgx_solver_t S, unsigned int *iterations, double *reduction_rate)
{
  MGX_REQUIRE(S, "mgx_solver_solve_cg_fused: null solver");
  const int      lmax = S->n_levels - 1;
  const size_t   n    = S->matrix[lmax]->d.n_dofs;
  mgx_context_t  ctx  = S->ctx;
  hipStream_t    s    = ctx->stream;
  mgx_operator_t A    = S->matrix_dp[lmax];
  double        *x = S->solution[lmax], *r = S->cg_r, *q = S->cg_z, *p = S->cg_d;
  MGX_REQUIRE(!ctx->has_comm, "mgx_solver_solve_cg_fused: single rank only");
  MGX_HIP(hipMemsetAsync(x, 0, 8 * n, s));
  MGX_HIP(hipMemsetAsync(p, 0, 8 * n, s));
  launch_copy_cast(s, r, MGX_F64, S->rhs[lmax], MGX_F64, n);
  double res0 = 0;
  MGX_TRY(mgx_operator_l2_norm(A, r, &res0));
  MGX_TRY(mgx_solver_vmult(S, q, r)); // q = z_0 = M r_0
  double rz = 0;
  MGX_TRY(dot(ctx, MGX_F64, r, q, n, &rz));
  double       alpha = 0., beta = 0., res = res0;
  unsigned int it = 0;
  S->cg_history.assign(1, res0);
  while (res > 1e-16 && res > 1e-9 * res0 && it < 1000) // ReductionControl(1000, 1e-16, 1e-9), :486
    {
      ++it;
      double sums[4];
      MGX_TRY(mgx_vmult_with_cg_update(A, alpha, beta, r, q, p, x, S->cg_h, sums));
      alpha = rz / sums[0];
      const uint32_t used = launch_axpy_norm(s, MGX_F64, r, q, -alpha, n, A->cg_partials);
      launch_reduce4(s, A->cg_partials, used, nullptr, A->cg_result);
      MGX_TRY(mgx_solver_vmult(S, q, r));
      double h[4], rz_new = 0;
      MGX_HIP(hipMemcpyAsync(h, A->cg_result, 4 * sizeof(double), hipMemcpyDeviceToHost, s));
      MGX_TRY(dot(ctx, MGX_F64, r, q, n, &rz_new)); // synchronises the stream
      res  = std::sqrt(h[2]);
      S->cg_history.push_back(res);
      beta = rz_new / rz;
      rz   = rz_new;
    }
  launch_sadd(s, MGX_F64, x, 1., alpha, p, n); // the update of x that the next step would have made
  if (iterations)
    *iterations = it;
  if (reduction_rate)
    *reduction_rate = it > 0 ? std::pow(res / res0, 1. / it) : 1.;
  MGX_HIP(hipGetLastError());
  if (it >= 1000)
    return fail(MGX_ERR_NOT_CONVERGED, "mgx_solver_solve_cg_fused: no convergence in 1000 iterations");
  return MGX_OK;
}

int mgx_solver_cg_history(mgx_solver_t S, double *history, int capacity, int *count)
{
  MGX_REQUIRE(S && count && (history || capacity == 0), "mgx_solver_cg_history: null argument");
  for (int i = 0; i < capacity && i < (int)S->cg_history.size(); ++i)
    history[i] = S->cg_history[i];
  *count = (int)S->cg_history.size();
  return MGX_OK;
}

int mgx_solver_do_matvec(mgx_solver_t S)
{
  MGX_REQUIRE(S, "mgx_solver_do_matvec: null solver");
  const int lmax = S->n_levels - 1;
  return mgx_vmult(S->matrix_dp[lmax], S->residual[lmax], S->solution[lmax]); // :627
}

int mgx_solver_do_matvec_smoother(mgx_solver_t S)
{
  MGX_REQUIRE(S, "mgx_solver_do_matvec_smoother: null solver");
  const int lmax = S->n_levels - 1;
  return mgx_vmult(S->matrix[lmax], S->solution_update[lmax], S->defect[lmax]); // :636
}

int mgx_solver_get_solution(mgx_solver_t S, int level, int insert_bc, const double **dptr)
{
  MGX_REQUIRE(S && dptr && level >= 0 && level < S->n_levels, "mgx_solver_get_solution: bad argument");
  if (insert_bc)
    set_bc(S, level, S->solution[level], false); // :379-380
  *dptr = S->solution[level];
  return MGX_OK;
}

int mgx_solver_get_vector(mgx_solver_t S, int level, int which, void **dptr)
{
  MGX_REQUIRE(S && dptr && level >= 0 && level < S->n_levels, "mgx_solver_get_vector: bad argument");
  switch (which)
    {
      case 0: *dptr = S->rhs[level]; break;
      case 1: *dptr = S->residual[level]; break;
      case 2: *dptr = S->defect[level]; break;
      case 3: *dptr = S->t[level]; break;
      case 4: *dptr = S->solution_update[level]; break;
      default: return fail(MGX_ERR_INVALID_ARGUMENT, "mgx_solver_get_vector: unknown vector id");
    }
  return MGX_OK;
}

// One V-cycle on the solver's own level vectors: defect[maxlevel] in, solution_update[maxlevel]
// out (mgx_solver_get_vector ids 2 and 4), for a caller that puts another level on top of the
// hierarchy (MultigridSolverDG::dg_v_cycle calls v_cycle(maxlevel, 1), multigrid_solver_dg.h:622)
int mgx_solver_v_cycle(mgx_solver_t S)
{
  MGX_REQUIRE(S, "mgx_solver_v_cycle: null solver");
  return v_cycle(S, S->n_levels - 1, 1);
}

// Re-creates the smoother of one level with other parameters (MultigridSolverDG sets up its FE_Q
// hierarchy with degree_pre - 1 on the finest level and a coarse tolerance of 2e-3,
// multigrid_solver_dg.h:271-291); degree < 0: from the tolerance, as on the coarsest level
int mgx_solver_reset_smoother(mgx_solver_t S, int level, double smoothing_range, int degree, int eig_cg_n_iterations)
{
  MGX_REQUIRE(S && level >= 0 && level < S->n_levels, "mgx_solver_reset_smoother: bad argument");
  MGX_REQUIRE(S->agg_solver == nullptr, "mgx_solver_reset_smoother: not on an agglomerated hierarchy");
  mgx_smoother_t sm = nullptr;
  MGX_TRY(mgx_smoother_create(S->matrix[level], smoothing_range, degree, eig_cg_n_iterations, &sm));
  MGX_TRY(mgx_smoother_destroy(S->smooth[level]));
  S->smooth[level] = sm;
  if (S->graph_exec) // the captured launches belong to the old smoother
    {
      MGX_HIP(hipStreamSynchronize(S->ctx->stream));
      (void)hipGraphExecDestroy(S->graph_exec);
      (void)hipGraphDestroy(S->graph);
      S->graph_exec  = nullptr;
      S->graph       = nullptr;
      S->graph_calls = 0;
    }
  return MGX_OK;
}

// level operators of a solver (V-cycle number type / fp64)
int mgx_solver_get_operator(mgx_solver_t S, int level, int fp64, mgx_operator_t *op)
{
  MGX_REQUIRE(S && op && level >= 0 && level < S->n_levels, "mgx_solver_get_operator: bad argument");
  *op = fp64 ? S->matrix_dp[level] : S->matrix[level];
  return MGX_OK;
}

int mgx_solver_n_levels(mgx_solver_t S) { return S ? S->n_levels : 0; }

// device view of an operator's compressed index table (the DG <-> FE_Q transfer of the DG level reads it)
int mgx_operator_device_indices(mgx_operator_t op, const uint32_t **idx27, uint32_t *n_cells, uint32_t *n_dofs, int *degree)
{
  MGX_REQUIRE(op, "mgx_operator_device_indices: null operator");
  if (idx27)
    *idx27 = op->d.idx27;
  if (n_cells)
    *n_cells = op->d.n_cells;
  if (n_dofs)
    *n_dofs = op->d.n_dofs;
  if (degree)
    *degree = op->d.p;
  return MGX_OK;
}

int mgx_solver_get_smoother(mgx_solver_t S, int level, mgx_smoother_t *sm)
{
  MGX_REQUIRE(S && sm && level >= 0 && level < S->n_levels, "mgx_solver_get_smoother: bad argument");
  *sm = S->smooth[level];
  return MGX_OK;
}

int mgx_solver_get_timings(mgx_solver_t S, double *timings)
{
  MGX_REQUIRE(S && timings, "mgx_solver_get_timings: null argument");
  std::copy(S->timings.begin(), S->timings.end(), timings);
  std::fill(S->timings.begin(), S->timings.end(), 0.); // print_wall_times resets :368-370
  return MGX_OK;
}

int mgx_solver_enable_timings(mgx_solver_t S, int enable)
{
  MGX_REQUIRE(S, "mgx_solver_enable_timings: null solver");
  S->timing = enable != 0;
  return MGX_OK;
}

} // extern "C"
