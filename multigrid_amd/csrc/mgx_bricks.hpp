// mgx_bricks.hpp -- host-side brick schedule builder (see mgx_bricks.cpp)
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace mgx
{
  struct BrickHost
  {
    uint32_t              n_bricks  = 0;
    int                   n_colours = 0;      // launch groups (one launch each)
    int                   n_iface_groups = 0; // leading groups made of the bricks on the rank interface (0: no split)
    int                   n_entities = 729; // entities per brick: 9^3 (4x4x4 cells) or 5^3 (2x2x2 cells)
    std::vector<uint32_t> colour_start; // [n_colours+1] into the colour-sorted brick order
    std::vector<uint32_t> ent_base;     // [n_bricks*729] first DoF per brick entity (constrained: invalid)
    std::vector<uint8_t>  ent_flags;    // [n_bricks*729] bit0 FIRST, bit1 LAST
    std::vector<uint32_t> order;        // [n_bricks] colour-sorted position -> brick index in cell order
  };

  // false (with a reason) if the level cannot be scheduled as 4x4x4 bricks
  // shared/n_shared: DoFs duplicated on other ranks (domain decomposition); their entities are
  // never flagged LAST because the sum is only complete after the interface exchange
  // split_interface: launch the bricks that touch a shared DoF first (see n_iface_groups)
  bool build_bricks(int p, uint32_t n_cells, uint32_t n_dofs, const uint32_t *idx27, const uint32_t *idx27_plain,
                    const uint8_t *colour_hint, const uint32_t *shared, uint32_t n_shared, bool split_interface,
                    BrickHost &out, std::string &why);

  // Reduced-colour schedule of the macro-element kernel (mgx_macro.hip, FREE; FreeSchedule in
  // mgx_internal.hpp).  n_classes = 1: one launch group, every entity on a brick surface is private;
  // n_classes = 2: bricks two-coloured over their faces (false if the face graph is not bipartite),
  // the face entities keep FIRST / LAST flags in the order of the launch groups, the entities on brick
  // edges and corners are private.  `bh` as returned by build_bricks (entity words without flags);
  // positions below refer to the group-sorted order of THIS schedule.
  struct FreeHost
  {
    int                   n_groups = 0, n_iface_groups = 0;
    std::vector<uint32_t> group_start;            // [n_groups + 1]
    std::vector<uint32_t> ent;                    // [n_bricks * n_entities]: first DoF | FIRST << 30 | LAST << 31, or invalid
    std::vector<uint32_t> surf_off;               // [n_entities]: offset of a private entity inside a block, or invalid
    uint32_t              n_surf = 0;             // private values per brick
    std::vector<uint32_t> surf_dof, surf_start, surf_pos; // private DoFs (shared with other ranks first) and their blocks' positions
    uint32_t              n_surf_shared = 0;
  };
  bool build_free_schedule(int p, const BrickHost &bh, uint32_t n_dofs, const uint32_t *shared, uint32_t n_shared,
                           bool split_interface, int n_classes, const std::vector<uint32_t> &item_map, FreeHost &out);

  // Item table of the macro-element kernel (mgx_macro.hip), one per degree: the (NB p + 1)^3 points
  // of a brick in the order they are gathered and written out.  Entities are taken cell after cell
  // in Morton order ({hex, x-face, y-face, xy-line, z-face, xz-line, yz-line, vertex} on the high
  // side / in the interior of each cell), then the entities on the three low faces of the brick.
  // First section: all pairs of DoFs (2k, 2k+1) of one entity -- adjacent in memory for every
  // numbering that satisfies the entity-contiguity contract, so one lane moves both with a 16-byte
  // access; second section: the last DoF of every entity of odd size.  Word: bits 0..9 entity
  // slot of the brick, 10..22 brick point (z G + y) G + x, 23..31 offset inside the entity.
  // MGX_MACRO_PAIRS=0 (default): no pair section, every DoF is a single item in entity order.
  // Measured on MI355X (135M DoFs, p = 4): the 16-byte accesses of the pair section buy nothing for
  // the plain form (107 vs 106 us per colour launch) and the scattered left-over singles cost the
  // fused Chebyshev forms 10-20 % (180 vs 150 us); kept as a build option for other numberings.
#ifndef MGX_MACRO_PAIRS
#define MGX_MACRO_PAIRS 0
#endif
  void build_item_map(int p, std::vector<uint32_t> &map);
  // Item table of the second pipeline of the macro-element kernel (mgx_macro2.hip): the same words, but the
  // (G - 2)^3 points in the interior of the brick first (in the order above), then the points on its surface.  Only
  // the surface points carry partial sums between the colour launches, can be constrained or lie on a rank
  // interface: the first (G - 2)^3 / threads value slots of every thread then need neither flags nor carrier accesses,
  // and the partial sums of the remaining few can be requested ahead of the sweeps.
  void build_item_map2(int p, std::vector<uint32_t> &map);
} // namespace mgx
