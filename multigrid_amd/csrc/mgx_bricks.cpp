// mgx_bricks.cpp -- host-side construction of the brick schedule consumed by mgx_brick.hip from
// the caller's 27-entry compressed index tables (laplace_operator.h:224-353).  Nothing here
// assumes the cube: the 4x4x4 structure of every group of 64 consecutive cells is *verified*
// through the entity indices the cells share, the brick adjacency is derived from shared surface
// entities, and a level that does not pass falls back to the per-cell kernel.
#include "mgx_bricks.hpp"

#include <omp.h>

#include <algorithm>
#include <array>
#include <cstring>
#include <parallel/algorithm>

namespace mgx
{
  namespace
  {
    constexpr uint32_t kInvalid    = 0xFFFFFFFFu;
    constexpr int      kMaxColours = 32;
    constexpr uint32_t kUnset      = 0xFFFFFFFEu;

    inline int compact3(int m)
    {
      return (m & 1) | ((m >> 2) & 2);
    }

    struct SurfaceRef
    {
      uint32_t key;   // first DoF of the entity (identifies it globally)
      uint32_t brick; // brick index (cell order)
      uint16_t slot;  // entity slot inside the brick
    };
  } // namespace

  bool build_bricks(int p, uint32_t n_cells, uint32_t n_dofs, const uint32_t *idx27, const uint32_t *idx27_plain,
                    const uint8_t *colour_hint, const uint32_t *shared, uint32_t n_shared, bool split_interface,
                    BrickHost &out, std::string &why)
  {
    out = BrickHost();
    if (p < 1 || p > 9)
      {
        why = "degree out of range";
        return false;
      }
    // 4x4x4 bricks for p <= 4, 2x2x2 (the children of one parent) for p >= 5
    const int NBd = p <= 4 ? 4 : 2, CB = NBd * NBd * NBd, E1 = 2 * NBd + 1, NE = E1 * E1 * E1;
    if (n_cells < (uint32_t)CB || n_cells % CB != 0)
      {
        why = "fewer cells than one brick or cell count not a multiple of the brick size";
        return false;
      }
    const uint32_t nb = n_cells / CB;
    out.n_entities    = NE;
    out.n_bricks      = nb;
    out.ent_base.assign((size_t)nb * NE, kInvalid);
    out.ent_flags.assign((size_t)nb * NE, 0);
    bool structured = true;
    // 1. entity table of every brick; consistency of the shared entities proves the 4x4x4 layout
#pragma omp parallel for schedule(static)
    for (uint32_t b = 0; b < nb; ++b)
      {
        uint32_t key[729];
        for (int i = 0; i < NE; ++i)
          key[i] = kUnset;
        uint32_t *ent = &out.ent_base[(size_t)b * NE];
        for (int m = 0; m < CB; ++m)
          {
            const int      bx = compact3(m), by = compact3(m >> 1), bz = compact3(m >> 2);
            const uint32_t c  = (uint32_t)CB * b + m;
            for (int e = 0; e < 27; ++e)
              {
                const int cx = e % 3, cy = (e / 3) % 3, cz = e / 9;
                const int size = (cx == 1 ? p - 1 : 1) * (cy == 1 ? p - 1 : 1) * (cz == 1 ? p - 1 : 1);
                const int slot = ((2 * bz + cz) * E1 + (2 * by + cy)) * E1 + 2 * bx + cx;
                if (size == 0)
                  {
                    key[slot] = kInvalid; // p = 1: lines/quads/hexes carry no DoFs
                    continue;
                  }
                const uint32_t v = idx27[27 * (size_t)c + e];
                const uint32_t k = idx27_plain ? idx27_plain[27 * (size_t)c + e] : v;
                if (key[slot] == kUnset)
                  {
                    key[slot] = k;
                    ent[slot] = v;
                  }
                else if (key[slot] != k || ent[slot] != v)
                  {
#pragma omp atomic write
                    structured = false;
                  }
              }
          }
      }
    if (!structured)
      {
        why = "consecutive cells do not form bricks in Morton order";
        out = BrickHost();
        return false;
      }
    // 2. references to the (unconstrained) surface entities; interior entities are complete
    //    after their own brick: FIRST|LAST
    std::vector<SurfaceRef> refs;
    {
      std::vector<std::vector<SurfaceRef>> local(omp_get_max_threads());
#pragma omp parallel for schedule(static)
      for (uint32_t b = 0; b < nb; ++b)
        {
          auto &mine = local[omp_get_thread_num()];
          for (int slot = 0; slot < NE; ++slot)
            {
              const uint32_t v = out.ent_base[(size_t)b * NE + slot];
              if (v == kInvalid)
                continue;
              const int ex = slot % E1, ey = (slot / E1) % E1, ez = slot / (E1 * E1);
              const bool surface = ex == 0 || ex == E1 - 1 || ey == 0 || ey == E1 - 1 || ez == 0 || ez == E1 - 1;
              if (surface)
                mine.push_back({v, b, (uint16_t)slot});
              else
                out.ent_flags[(size_t)b * NE + slot] = 3;
            }
        }
      size_t total = 0;
      for (auto &l : local)
        total += l.size();
      refs.reserve(total);
      for (auto &l : local)
        refs.insert(refs.end(), l.begin(), l.end());
    }
    __gnu_parallel::sort(refs.begin(), refs.end(), [](const SurfaceRef &a, const SurfaceRef &b) {
      return a.key != b.key ? a.key < b.key : a.brick < b.brick;
    });
    // 3. colours: bricks sharing an entity must differ
    std::vector<uint8_t> colour(nb, 255);
    if (colour_hint)
      {
        for (uint32_t b = 0; b < nb; ++b)
          colour[b] = colour_hint[b];
      }
    else
      {
        // adjacency from the groups, greedy colouring in brick order
        std::vector<std::vector<uint32_t>> adj(nb);
        for (size_t i = 0; i < refs.size();)
          {
            size_t j = i;
            while (j < refs.size() && refs[j].key == refs[i].key)
              ++j;
            for (size_t a = i; a < j; ++a)
              for (size_t c = i; c < j; ++c)
                if (a != c)
                  adj[refs[a].brick].push_back(refs[c].brick);
            i = j;
          }
        for (uint32_t b = 0; b < nb; ++b)
          {
            uint64_t used = 0;
            for (uint32_t o : adj[b])
              if (colour[o] != 255)
                used |= 1ull << colour[o];
            int c = 0;
            while (used & (1ull << c))
              ++c;
            if (c >= kMaxColours)
              {
                why = "more than 32 brick colours needed";
                out = BrickHost();
                return false;
              }
            colour[b] = (uint8_t)c;
          }
      }
    int n_colours = 0;
    for (uint32_t b = 0; b < nb; ++b)
      {
        if (colour[b] >= kMaxColours)
          {
            why = "brick colour out of range";
            out = BrickHost();
            return false;
          }
        n_colours = std::max(n_colours, colour[b] + 1);
      }
    // 3b. launch groups.  Normally one per colour.  On a decomposed mesh the bricks that touch an
    //     interface DoF can be launched first (groups 0 .. n_colours-1, colour by colour), so that
    //     the exchange of the interface sums overlaps with the interior bricks (groups n_colours ..
    //     2 n_colours-1); the FIRST / LAST flags below follow the group order.
    std::vector<uint8_t> group(colour);
    int                  n_groups = n_colours;
    std::vector<uint8_t> is_shared;
    if (n_shared > 0)
      {
        is_shared.assign(n_dofs, 0);
        for (uint32_t i = 0; i < n_shared; ++i)
          is_shared[shared[i]] = 1;
      }
    if (split_interface && n_shared > 0 && 2 * n_colours <= kMaxColours)
      {
        uint32_t n_iface = 0;
#pragma omp parallel for schedule(static) reduction(+ : n_iface)
        for (uint32_t b = 0; b < nb; ++b)
          {
            bool iface = false;
            for (int slot = 0; slot < NE && !iface; ++slot)
              {
                const uint32_t v = out.ent_base[(size_t)b * NE + slot];
                iface            = v != kInvalid && is_shared[v];
              }
            if (!iface)
              group[b] = (uint8_t)(colour[b] + n_colours);
            else
              ++n_iface;
          }
        if (n_iface > 0 && n_iface < nb)
          {
            n_groups          = 2 * n_colours;
            out.n_iface_groups = n_colours;
          }
        else
          group = colour;
      }
    // 4. FIRST / LAST flags of the surface entities from the launch order (= group order)
    for (size_t i = 0; i < refs.size();)
      {
        size_t j = i;
        while (j < refs.size() && refs[j].key == refs[i].key)
          ++j;
        size_t lo = i, hi = i;
        for (size_t a = i; a < j; ++a)
          {
            for (size_t c = a + 1; c < j; ++c)
              if (colour[refs[a].brick] == colour[refs[c].brick])
                {
                  why = "two bricks that share DoFs have the same colour";
                  out = BrickHost();
                  return false;
                }
            if (group[refs[a].brick] < group[refs[lo].brick])
              lo = a;
            if (group[refs[a].brick] > group[refs[hi].brick])
              hi = a;
          }
        out.ent_flags[(size_t)refs[lo].brick * NE + refs[lo].slot] |= 1;
        out.ent_flags[(size_t)refs[hi].brick * NE + refs[hi].slot] |= 2;
        i = j;
      }
    // 4b. interface entities of a decomposed mesh are complete only after the exchange
    if (n_shared > 0)
      {
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < out.ent_base.size(); ++i)
          if (out.ent_base[i] != kInvalid && is_shared[out.ent_base[i]])
            out.ent_flags[i] &= (uint8_t)~2u;
      }
    // 5. sort the bricks by launch group (stable in cell order) and permute the tables
    std::vector<uint32_t> order(nb);
    out.colour_start.assign(n_groups + 1, 0);
    for (uint32_t b = 0; b < nb; ++b)
      out.colour_start[group[b] + 1]++;
    for (int c = 0; c < n_groups; ++c)
      out.colour_start[c + 1] += out.colour_start[c];
    {
      std::vector<uint32_t> pos(out.colour_start.begin(), out.colour_start.end() - 1);
      for (uint32_t b = 0; b < nb; ++b)
        order[pos[group[b]]++] = b;
    }
    std::vector<uint32_t> base2((size_t)nb * NE);
    std::vector<uint8_t>  flags2((size_t)nb * NE);
#pragma omp parallel for schedule(static)
    for (uint32_t i = 0; i < nb; ++i)
      {
        std::memcpy(&base2[(size_t)i * NE], &out.ent_base[(size_t)order[i] * NE], sizeof(uint32_t) * NE);
        std::memcpy(&flags2[(size_t)i * NE], &out.ent_flags[(size_t)order[i] * NE], NE);
      }
    out.ent_base.swap(base2);
    out.ent_flags.swap(flags2);
    out.order = order;
    out.n_colours = n_groups;
    (void)n_dofs;
    return true;
  }

  bool build_free_schedule(int p, const BrickHost &bh, uint32_t n_dofs, const uint32_t *shared, uint32_t n_shared,
                           bool split_interface, int n_classes, const std::vector<uint32_t> &item_map, FreeHost &out)
  {
    out = FreeHost();
    const int      NBd = p <= 4 ? 4 : 2, E1 = 2 * NBd + 1, NE = E1 * E1 * E1;
    const uint32_t nb  = bh.n_bricks;
    if (nb == 0 || bh.n_entities != NE || (n_classes != 1 && n_classes != 2))
      return false;
    auto coords = [&](int slot, int(&e)[3]) {
      e[0] = slot % E1;
      e[1] = (slot / E1) % E1;
      e[2] = slot / (E1 * E1);
    };
    auto n_boundary = [&](int slot) {
      int e[3];
      coords(slot, e);
      return (e[0] == 0 || e[0] == E1 - 1) + (e[1] == 0 || e[1] == E1 - 1) + (e[2] == 0 || e[2] == E1 - 1);
    };
    auto ent_size = [&](int slot) {
      int e[3];
      coords(slot, e);
      return (uint32_t)(((e[0] & 1) ? p - 1 : 1) * ((e[1] & 1) ? p - 1 : 1) * ((e[2] & 1) ? p - 1 : 1));
    };
    // face f = 2 axis + side of a surface entity with exactly one boundary coordinate
    auto face_of = [&](int slot) {
      int e[3];
      coords(slot, e);
      for (int a = 0; a < 3; ++a)
        if (e[a] == 0 || e[a] == E1 - 1)
          return 2 * a + (e[a] == 0 ? 0 : 1);
      return -1;
    };
    // 1. the brick across each face, through a vertex in the interior of the face
    std::vector<uint32_t> partner((size_t)nb * 6, kInvalid);
    {
      std::vector<std::pair<uint32_t, uint32_t>> keys; // (first DoF of the vertex, 6 brick + face)
      keys.reserve((size_t)nb * 6);
      for (uint32_t b = 0; b < nb; ++b)
        for (int f = 0; f < 6; ++f)
          {
            int e[3] = {2, 2, 2};
            e[f / 2]  = (f & 1) ? E1 - 1 : 0;
            const uint32_t v = bh.ent_base[(size_t)b * NE + (e[2] * E1 + e[1]) * E1 + e[0]];
            if (v != kInvalid)
              keys.push_back({v & 0x3FFFFFFFu, 6 * b + (uint32_t)f});
          }
      std::sort(keys.begin(), keys.end());
      for (size_t i = 0; i + 1 < keys.size(); ++i)
        if (keys[i].first == keys[i + 1].first)
          {
            if (i + 2 < keys.size() && keys[i + 2].first == keys[i].first)
              return false; // a face vertex in three bricks: not a brick mesh
            partner[keys[i].second]     = keys[i + 1].second / 6;
            partner[keys[i + 1].second] = keys[i].second / 6;
          }
    }
    // 2. classes: two-colouring over the faces (breadth first), or one class
    std::vector<uint8_t> cls(nb, n_classes == 1 ? 0 : 255);
    if (n_classes == 2)
      {
        std::vector<uint32_t> queue;
        for (uint32_t seed = 0; seed < nb; ++seed)
          {
            if (cls[seed] != 255)
              continue;
            cls[seed] = 0;
            queue.assign(1, seed);
            for (size_t q = 0; q < queue.size(); ++q)
              {
                const uint32_t b = queue[q];
                for (int f = 0; f < 6; ++f)
                  {
                    const uint32_t o = partner[(size_t)b * 6 + f];
                    if (o == kInvalid)
                      continue;
                    if (cls[o] == 255)
                      {
                        cls[o] = (uint8_t)(1 - cls[b]);
                        queue.push_back(o);
                      }
                    else if (cls[o] == cls[b])
                      return false; // not bipartite
                  }
              }
          }
      }
    // 3. launch groups: class, interface bricks first on a decomposed mesh
    std::vector<uint8_t> is_shared;
    if (n_shared > 0)
      {
        is_shared.assign(n_dofs, 0);
        for (uint32_t i = 0; i < n_shared; ++i)
          is_shared[shared[i]] = 1;
      }
    std::vector<uint8_t> group(cls);
    int                  n_groups = n_classes;
    if (split_interface && n_shared > 0)
      {
        uint32_t n_iface = 0;
        for (uint32_t b = 0; b < nb; ++b)
          {
            bool iface = false;
            for (int slot = 0; slot < NE && !iface; ++slot)
              {
                const uint32_t v = bh.ent_base[(size_t)b * NE + slot];
                iface            = v != kInvalid && is_shared[v & 0x3FFFFFFFu];
              }
            if (!iface)
              group[b] = (uint8_t)(cls[b] + n_classes);
            else
              ++n_iface;
          }
        if (n_iface > 0 && n_iface < nb)
          {
            n_groups           = 2 * n_classes;
            out.n_iface_groups = n_classes;
          }
        else
          group = cls;
      }
    out.n_groups = n_groups;
    out.group_start.assign(n_groups + 1, 0);
    for (uint32_t b = 0; b < nb; ++b)
      out.group_start[group[b] + 1]++;
    for (int g = 0; g < n_groups; ++g)
      out.group_start[g + 1] += out.group_start[g];
    std::vector<uint32_t> order(nb);
    {
      std::vector<uint32_t> pos(out.group_start.begin(), out.group_start.end() - 1);
      for (uint32_t b = 0; b < nb; ++b)
        order[pos[group[b]]++] = b;
    }
    // 4. private entities and their offsets inside a block, in item order (consecutive items of the
    //    write-out then store to consecutive addresses)
    out.surf_off.assign(NE, kInvalid);
    for (const uint32_t m : item_map)
      {
        const int slot = (int)(m & 1023u);
        const int nbd  = n_boundary(slot);
        if ((n_classes == 1 ? nbd >= 1 : nbd >= 2) && out.surf_off[slot] == kInvalid && ent_size(slot) > 0)
          {
            out.surf_off[slot] = out.n_surf;
            out.n_surf += ent_size(slot);
          }
      }
    // 5. entity table in group order with the flags of this schedule
    out.ent.assign((size_t)nb * NE, kInvalid);
#pragma omp parallel for schedule(static)
    for (uint32_t i = 0; i < nb; ++i)
      {
        const uint32_t b = order[i];
        for (int slot = 0; slot < NE; ++slot)
          {
            const uint32_t v = bh.ent_base[(size_t)b * NE + slot];
            if (v == kInvalid)
              continue;
            const uint32_t idx = v & 0x3FFFFFFFu;
            const int      nbd = n_boundary(slot);
            uint32_t       flags = 3u; // interior: FIRST | LAST
            if (out.surf_off[slot] != kInvalid)
              flags = 1u; // private
            else if (nbd == 1)
              {
                const uint32_t o = partner[(size_t)b * 6 + face_of(slot)];
                if (o != kInvalid)
                  flags = group[b] < group[o] ? 1u : 2u;
              }
            if (!is_shared.empty() && is_shared[idx])
              flags &= ~2u; // complete only after the exchange
            out.ent[(size_t)i * NE + slot] = idx | (flags << 30);
          }
      }
    // 6. private DoFs -> positions of their partial sums, ascending (the fixed order of the sum);
    //    counting sort by DoF; the DoFs shared with other ranks first
    if (out.n_surf > 0)
      {
        auto for_each_ref = [&](auto &&f) {
          for (uint32_t i = 0; i < nb; ++i)
            for (int slot = 0; slot < NE; ++slot)
              {
                const uint32_t w = out.ent[(size_t)i * NE + slot];
                if (out.surf_off[slot] == kInvalid || w == kInvalid)
                  continue;
                for (uint32_t o = 0; o < ent_size(slot); ++o)
                  f((w & 0x3FFFFFFFu) + o, i * out.n_surf + out.surf_off[slot] + o);
              }
        };
        std::vector<uint32_t> cnt((size_t)n_dofs + 1, 0);
        for_each_ref([&](uint32_t dof, uint32_t) { ++cnt[dof + 1]; });
        std::vector<uint32_t> slot_of((size_t)n_dofs, kInvalid);
        out.surf_start.assign(1, 0);
        for (int pass = 0; pass < 2; ++pass)
          {
            for (uint32_t dof = 0; dof < n_dofs; ++dof)
              if (cnt[dof + 1] > 0 && (!is_shared.empty() && is_shared[dof]) == (pass == 0))
                {
                  slot_of[dof] = (uint32_t)out.surf_dof.size();
                  out.surf_dof.push_back(dof);
                  out.surf_start.push_back(out.surf_start.back() + cnt[dof + 1]);
                }
            if (pass == 0)
              out.n_surf_shared = (uint32_t)out.surf_dof.size();
          }
        out.surf_pos.assign((size_t)out.surf_start.back(), 0);
        std::vector<uint32_t> fill(out.surf_start.begin(), out.surf_start.end() - 1);
        for_each_ref([&](uint32_t dof, uint32_t pos) { out.surf_pos[fill[slot_of[dof]]++] = pos; });
      }
    else
      out.surf_start.assign(1, 0);
    return true;
  }

  namespace
  {
    // the entities of a brick in the write-out order of build_item_map with the brick point of each of their DoFs
    struct ItemEntity
    {
      int              slot;
      std::vector<int> pnt; // brick point of DoF k of the entity
    };
    inline uint32_t item_word(int slot, int pnt, int off)
    {
      return (uint32_t)slot | ((uint32_t)pnt << 10) | ((uint32_t)off << 23);
    }
    void enumerate_item_entities(int p, std::vector<ItemEntity> &ents)
    {
      const int NB = p <= 4 ? 4 : 2, G = NB * p + 1, E1 = 2 * NB + 1;
      ents.clear();
      // part A: per cell (Morton order) the entities on its high side / in its interior
      for (int m = 0; m < NB * NB * NB; ++m)
        {
          const int bx = compact3(m), by = compact3(m >> 1), bz = compact3(m >> 2);
          for (int j = 0; j < 8; ++j)
            {
              const int  cx = 1 + (j & 1), cy = 1 + ((j >> 1) & 1), cz = 1 + (j >> 2);
              const int  nx = cx == 1 ? p - 1 : 1, ny = cy == 1 ? p - 1 : 1, nz = cz == 1 ? p - 1 : 1;
              ItemEntity e;
              e.slot = ((2 * bz + cz) * E1 + 2 * by + cy) * E1 + 2 * bx + cx;
              for (int kl = 0; kl < nx * ny * nz; ++kl)
                {
                  const int ox = kl % nx, oy = (kl / nx) % ny, oz = kl / (nx * ny);
                  const int lx = cx == 1 ? 1 + ox : p, ly = cy == 1 ? 1 + oy : p, lz = cz == 1 ? 1 + oz : p;
                  e.pnt.push_back(((bz * p + lz) * G + by * p + ly) * G + bx * p + lx);
                }
              if (!e.pnt.empty())
                ents.push_back(e);
            }
        }
      // part B: the entities on the three low faces of the brick (their DoFs belong to the cell
      // blocks of neighbouring bricks), plane by plane
      for (int ez = 0; ez < E1; ++ez)
        for (int ey = 0; ey < E1; ++ey)
          for (int ex = 0; ex < E1; ++ex)
            {
              if (ex != 0 && ey != 0 && ez != 0)
                continue;
              const int  nx = (ex & 1) ? p - 1 : 1, ny = (ey & 1) ? p - 1 : 1, nz = (ez & 1) ? p - 1 : 1;
              ItemEntity e;
              e.slot = (ez * E1 + ey) * E1 + ex;
              for (int kl = 0; kl < nx * ny * nz; ++kl)
                {
                  const int ox = kl % nx, oy = (kl / nx) % ny, oz = kl / (nx * ny);
                  const int gx = (ex / 2) * p + ((ex & 1) ? 1 + ox : 0), gy = (ey / 2) * p + ((ey & 1) ? 1 + oy : 0),
                            gz = (ez / 2) * p + ((ez & 1) ? 1 + oz : 0);
                  e.pnt.push_back((gz * G + gy) * G + gx);
                }
              if (!e.pnt.empty())
                ents.push_back(e);
            }
    }
  } // namespace

  void build_item_map(int p, std::vector<uint32_t> &map)
  {
    const int               NB = p <= 4 ? 4 : 2, G = NB * p + 1;
    std::vector<ItemEntity> ents;
    enumerate_item_entities(p, ents);
    // pairs first (DoFs 2k, 2k+1 of one entity: adjacent in memory by the entity-contiguity
    // contract, whatever the numbering), then the odd DoF left over in every entity of odd size
    map.clear();
    map.reserve((size_t)G * G * G);
    if (MGX_MACRO_PAIRS)
      {
        for (const ItemEntity &e : ents)
          for (size_t k = 0; k + 1 < e.pnt.size(); k += 2)
            {
              map.push_back(item_word(e.slot, e.pnt[k], (int)k));
              map.push_back(item_word(e.slot, e.pnt[k + 1], (int)k + 1));
            }
        for (const ItemEntity &e : ents)
          if (e.pnt.size() % 2)
            map.push_back(item_word(e.slot, e.pnt.back(), (int)e.pnt.size() - 1));
      }
    else
      for (const ItemEntity &e : ents)
        for (size_t k = 0; k < e.pnt.size(); ++k)
          map.push_back(item_word(e.slot, e.pnt[k], (int)k));
  }

  void build_item_map2(int p, std::vector<uint32_t> &map)
  {
    const int               NB = p <= 4 ? 4 : 2, G = NB * p + 1, E1 = 2 * NB + 1;
    std::vector<ItemEntity> ents;
    enumerate_item_entities(p, ents);
    auto on_surface = [&](int slot) {
      const int ex = slot % E1, ey = (slot / E1) % E1, ez = slot / (E1 * E1);
      return ex == 0 || ex == E1 - 1 || ey == 0 || ey == E1 - 1 || ez == 0 || ez == E1 - 1;
    };
    map.clear();
    map.reserve((size_t)G * G * G);
    for (int pass = 0; pass < 2; ++pass)
      for (const ItemEntity &e : ents)
        if ((int)on_surface(e.slot) == pass)
          for (size_t k = 0; k < e.pnt.size(); ++k)
            map.push_back(item_word(e.slot, e.pnt[k], (int)k));
  }
} // namespace mgx
