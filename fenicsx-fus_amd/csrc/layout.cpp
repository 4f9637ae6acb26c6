// layout.cpp -- block partition, conflict-free rounds and internal DOF numbering (host only).
// See layout.hpp.  Replaces the role of the reference's serial cell order + global scatter-add
// (cpp/fenicsx-sf/common/spectral_op.hpp:183-242) with a structure the GPU can execute without
// atomics.
#include "layout.hpp"

#include <algorithm>
#include <cstring>
#include <map>
#include <numeric>
#include <unordered_map>

namespace fus
{
namespace
{
struct Rcb
{
  const double* cen;
  std::vector<int32_t>& order;
  std::vector<std::pair<int64_t, int64_t>>& leaves;

  void split(int64_t lo, int64_t hi, int64_t nparts)
  {
    if (nparts <= 1 || hi - lo <= 1)
    {
      leaves.emplace_back(lo, hi);
      return;
    }
    double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
    for (int64_t k = lo; k < hi; ++k)
      for (int d = 0; d < 3; ++d)
      {
        const double v = cen[3 * (int64_t)order[k] + d];
        mn[d] = std::min(mn[d], v);
        mx[d] = std::max(mx[d], v);
      }
    int ax = 0;
    for (int d = 1; d < 3; ++d)
      if (mx[d] - mn[d] > (mx[ax] - mn[ax]) * (1.0 + 1e-9))
        ax = d;
    const int64_t nl = nparts / 2;
    const int64_t mid = lo + ((hi - lo) * nl) / nparts;
    const double* c = cen;
    std::nth_element(order.begin() + lo, order.begin() + mid, order.begin() + hi,
                     [c, ax](int32_t a, int32_t b)
                     {
                       const double va = c[3 * (int64_t)a + ax], vb = c[3 * (int64_t)b + ax];
                       return va < vb || (va == vb && a < b);
                     });
    split(lo, mid, nl);
    split(mid, hi, nparts - nl);
  }
};
} // namespace

std::string build_layout(Layout& L, int P, int64_t ncells, int64_t ndofs,
                         const int32_t* dm, const double* centroids, int block_elems, int waves,
                         const uint8_t* force_shared, int tdim, int slot_factor)
{
  if (P < 1 || P > 15)
    return "unsupported degree";
  if (tdim != 2 && tdim != 3)
    return "unsupported topological dimension";
  if (ncells <= 0 || ndofs <= 0)
    return "empty mesh";
  L = Layout();
  L.P = P, L.N = P + 1, L.tdim = tdim, L.Nd = tdim == 3 ? L.N * L.N * L.N : L.N * L.N;
  L.ncells = ncells, L.ndofs = ndofs;
  L.waves = std::max(1, waves);
  L.epw = std::max(1, 64 / (L.N * L.N));
  // degrees 8-10: a tensor plane (an element, for quadrilaterals) has more than 64 columns and two waves share an element
  L.slots = (L.N * L.N > 64) ? std::max(1, L.waves / 2) : L.waves * L.epw;
  L.slots *= std::max(1, slot_factor);   // packed fp32 kernels: two elements per lane group and trip
  const int Nd = L.Nd;
  if (block_elems < 1)
    block_elems = 64;

  // ---- 1. recursive coordinate bisection into compact blocks ----
  std::vector<int32_t> order(ncells);
  std::iota(order.begin(), order.end(), 0);
  std::vector<std::pair<int64_t, int64_t>> leaves;
  const int64_t nparts = (ncells + block_elems - 1) / block_elems;
  Rcb{centroids, order, leaves}.split(0, ncells, nparts);
  L.nblocks = (int32_t)leaves.size();
  for (auto& lf : leaves)
    std::sort(order.begin() + lf.first, order.begin() + lf.second);

  // blocks touching a dof other ranks hold as well go first: the operator launches them ahead of
  // the rest so that the interface exchange overlaps the remaining blocks (fusmi.hip stage_begin)
  L.nblocks_if = 0;
  if (force_shared)
  {
    auto touches = [&](const std::pair<int64_t, int64_t>& lf)
    {
      for (int64_t k = lf.first; k < lf.second; ++k)
      {
        const int32_t* d = dm + (int64_t)order[k] * Nd;
        for (int i = 0; i < Nd; ++i)
          if (d[i] >= 0 && d[i] < ndofs && force_shared[d[i]])
            return true;
      }
      return false;
    };
    auto mid = std::stable_partition(leaves.begin(), leaves.end(), touches);
    L.nblocks_if = (int32_t)(mid - leaves.begin());
  }

  // ---- 2. how many blocks touch each dof ----
  std::vector<int32_t> last_blk(ndofs, -1);
  std::vector<uint8_t> nblk(ndofs, 0);
  std::vector<uint64_t> sharers(ndofs, 0);  // signature of the set of blocks touching the dof (its "group": a face,
                                            // an edge or a corner between blocks)
  for (int32_t b = 0; b < L.nblocks; ++b)
    for (int64_t k = leaves[b].first; k < leaves[b].second; ++k)
    {
      const int32_t* d = dm + (int64_t)order[k] * Nd;
      for (int i = 0; i < Nd; ++i)
      {
        if (d[i] < 0 || d[i] >= ndofs)
          return "dofmap entry out of range";
        if (last_blk[d[i]] != b)
        {
          last_blk[d[i]] = b;
          if (nblk[d[i]] < 255)
            ++nblk[d[i]];
          sharers[d[i]] = (sharers[d[i]] ^ (uint64_t)(b + 1)) * 0x9E3779B97F4A7C15ull;
        }
      }
    }

  // dofs other ranks also hold are never complete inside one block of this rank
  if (force_shared)
    for (int64_t g = 0; g < ndofs; ++g)
      if (force_shared[g] && nblk[g] == 1)
        nblk[g] = 2;

  // ---- 3. per block: rounds, local numbering, internal numbering ----
  L.cell_perm.resize(ncells);
  L.blk_elem_off.assign(L.nblocks + 1, 0);
  L.blk_shape.resize(L.nblocks);
  L.blk_int_off.resize(L.nblocks);
  L.blk_sh_off.assign(L.nblocks + 1, 0);
  L.dof_perm.assign(ndofs, -1);
  std::vector<int32_t> sh_id(ndofs, -1);          // shared dof -> shared index
  std::vector<int32_t> loc_of(ndofs, -1), loc_blk(ndofs, -1);
  std::vector<int32_t> pair_shid;                 // [npairs] shared index of each pair
  std::map<std::vector<uint16_t>, int32_t> shape_map;
  int64_t int_cursor = 0;
  std::vector<uint64_t> rmask;
  std::vector<uint16_t> key;
  std::vector<int32_t> shl;                          // the block's shared dofs
  std::unordered_map<uint64_t, int32_t> grp_rank;    // group signature -> order of first appearance in the block

  for (int32_t b = 0; b < L.nblocks; ++b)
  {
    const int64_t lo = leaves[b].first, hi = leaves[b].second;
    const int32_t nelem = (int32_t)(hi - lo);
    L.blk_elem_off[b + 1] = L.blk_elem_off[b] + nelem;
    if (nelem > 32767)
      return "block too large";

    // temporary local ids (first appearance) for the conflict masks
    int32_t ntmp = 0;
    for (int64_t k = lo; k < hi; ++k)
    {
      const int32_t* d = dm + (int64_t)order[k] * Nd;
      for (int i = 0; i < Nd; ++i)
        if (loc_blk[d[i]] != b)
          loc_blk[d[i]] = b, loc_of[d[i]] = ntmp++;
    }
    if (ntmp > 65535)
      return "block has more than 65535 local dofs";
    rmask.assign(ntmp, 0);
    // greedy first-fit: an element goes to the first round where it shares no dof with the
    // round's other elements and a slot is free
    std::vector<std::vector<int32_t>> rd;  // rd[r] = positions (k - lo)
    for (int64_t k = lo; k < hi; ++k)
    {
      const int32_t* d = dm + (int64_t)order[k] * Nd;
      uint64_t busy = 0;
      for (int i = 0; i < Nd; ++i)
        busy |= rmask[loc_of[d[i]]];
      int r = 0;
      for (;; ++r)
      {
        if (r >= 64)
          return "block needs more than 64 rounds";
        if (busy & (1ull << r))
          continue;
        if (r < (int)rd.size() && (int)rd[r].size() >= L.slots)
          continue;
        break;
      }
      if (r >= (int)rd.size())
        rd.resize(r + 1);
      rd[r].push_back((int32_t)(k - lo));
      for (int i = 0; i < Nd; ++i)
        rmask[loc_of[d[i]]] |= (1ull << r);
    }
    const int32_t nrounds = (int32_t)rd.size();
    L.max_rounds = std::max(L.max_rounds, nrounds);
    L.max_nelem = std::max(L.max_nelem, nelem);

    // internal element order = (round, slot), compact
    std::vector<int16_t> rtab((size_t)nrounds * L.slots, (int16_t)-1);
    std::vector<int32_t> erel_cell(nelem);
    int32_t er = 0;
    for (int r = 0; r < nrounds; ++r)
      for (size_t s = 0; s < rd[r].size(); ++s)
      {
        rtab[(size_t)r * L.slots + s] = (int16_t)er;
        erel_cell[er] = order[lo + rd[r][s]];
        L.cell_perm[L.blk_elem_off[b] + er] = erel_cell[er];
        ++er;
      }

    // final local numbering: interior dofs first (by first appearance), then shared, group after group (groups
    // and the dofs of a group by first appearance): the dofs a block shares with one set of other blocks are
    // then a run of its slots AND, numbered below in this order, a run of the shared range -- the gather of
    // their values and the store of their partial sums touch whole cache lines.  Blocks of one shape still get
    // one local dofmap (the order depends on the block's own traversal only).
    int32_t nint = 0, nsh = 0;
    shl.clear();
    for (int32_t e = 0; e < nelem; ++e)
    {
      const int32_t* d = dm + (int64_t)erel_cell[e] * Nd;
      for (int i = 0; i < Nd; ++i)
      {
        const int32_t g = d[i];
        if (loc_blk[g] != b)
          continue;
        loc_blk[g] = -2 - b;   // mark assigned
        if (nblk[g] > 1)
          shl.push_back(g);
        else
          loc_of[g] = nint++;
      }
    }
    {
      grp_rank.clear();
      for (int32_t g : shl)
        grp_rank.emplace(sharers[g], (int32_t)grp_rank.size());
      std::stable_sort(shl.begin(), shl.end(),
                       [&](int32_t a, int32_t c) { return grp_rank[sharers[a]] < grp_rank[sharers[c]]; });
      for (int32_t g : shl)
      {
        loc_of[g] = nint + nsh++;
        if (sh_id[g] < 0)
          sh_id[g] = (int32_t)L.n_shared++;
      }
    }
    const int32_t nloc = nint + nsh;
    L.max_nloc = std::max(L.max_nloc, nloc);

    // internal numbering
    int_cursor = (int_cursor + 15) & ~(int64_t)15;
    if (int_cursor + nint > 2000000000ll)
      return "local vector exceeds int32 indexing";
    L.blk_int_off[b] = (int32_t)int_cursor;
    L.blk_sh_off[b + 1] = L.blk_sh_off[b] + nsh;
    L.sh_gidx.resize(L.blk_sh_off[b + 1]);
    pair_shid.resize(L.blk_sh_off[b + 1]);

    key.clear();
    key.push_back((uint16_t)nelem), key.push_back((uint16_t)nloc);
    key.push_back((uint16_t)nint), key.push_back((uint16_t)nrounds);
    for (auto v : rtab)
      key.push_back((uint16_t)v);
    const size_t ldm_start = key.size();
    key.resize(ldm_start + (size_t)nelem * Nd);
    for (int32_t e = 0; e < nelem; ++e)
    {
      const int32_t* d = dm + (int64_t)erel_cell[e] * Nd;
      for (int i = 0; i < Nd; ++i)
      {
        const int32_t g = d[i];
        const int32_t l = loc_of[g];
        key[ldm_start + (size_t)e * Nd + i] = (uint16_t)l;
        if (l < nint)
          L.dof_perm[g] = (int32_t)(int_cursor + l);
        else
          pair_shid[L.blk_sh_off[b] + (l - nint)] = sh_id[g];
      }
    }
    int_cursor += nint;
    L.n_interior += nint;

    auto it = shape_map.find(key);
    if (it == shape_map.end())
    {
      Layout::Shape sh;
      sh.nelem = nelem, sh.nloc = nloc, sh.nint = nint, sh.nrounds = nrounds;
      sh.rounds_off = (int64_t)L.rounds.size();
      L.rounds.insert(L.rounds.end(), rtab.begin(), rtab.end());
      L.ldm.resize((L.ldm.size() + 7) & ~(size_t)7);
      sh.ldm_off = (int64_t)L.ldm.size();
      L.ldm.insert(L.ldm.end(), key.begin() + ldm_start, key.end());
      const int32_t id = (int32_t)L.shapes.size();
      L.shapes.push_back(sh);
      it = shape_map.emplace(key, id).first;
    }
    L.blk_shape[b] = it->second;
  }

  // dofs no cell touches (none in a consistent mesh) still need a slot
  for (int64_t g = 0; g < ndofs; ++g)
    if (nblk[g] == 0)
      sh_id[g] = (int32_t)L.n_shared++;

  // Order of the shared range.  Rank-local shared dofs first, by DESCENDING number of sharing blocks (first
  // appearance within a class): the dofs with more than k sharers are then a prefix of the range for every k,
  // which is what lets the partial sums be stored as planes (below).  Shared dofs held by other ranks too
  // (interface dofs) go last, so that the rank-local shared dofs and the interface dofs are two contiguous
  // index ranges; the interface range starts on a 128-byte boundary (the slots in between stay empty).
  L.n_shared_local = L.n_shared;
  L.n_if_start_pad = L.n_shared;
  {
    std::vector<int64_t> owner(L.n_shared, -1);
    for (int64_t g = 0; g < ndofs; ++g)
      if (sh_id[g] >= 0)
        owner[sh_id[g]] = g;
    std::vector<int32_t> mult(L.n_shared, 0);
    for (auto v : pair_shid)
      ++mult[v];
    std::vector<int32_t> local;
    local.reserve(L.n_shared);
    for (int64_t k = 0; k < L.n_shared; ++k)
      if (!(force_shared && force_shared[owner[k]]))
        local.push_back((int32_t)k);
    std::stable_sort(local.begin(), local.end(), [&](int32_t a, int32_t b) { return mult[a] > mult[b]; });
    const int64_t nlocal = (int64_t)local.size(), nif = L.n_shared - nlocal;
    const int64_t start = nif > 0 ? ((nlocal + 15) & ~(int64_t)15) : nlocal;
    std::vector<int32_t> remap(L.n_shared, -1);
    for (int64_t k = 0; k < nlocal; ++k)
      remap[local[k]] = (int32_t)k;
    int64_t ci = start;
    for (int64_t k = 0; k < L.n_shared; ++k)
      if (remap[k] < 0)
        remap[k] = (int32_t)ci++;
    for (int64_t g = 0; g < ndofs; ++g)
      if (sh_id[g] >= 0)
        sh_id[g] = remap[sh_id[g]];
    for (auto& v : pair_shid)
      v = remap[v];
    L.n_shared_local = nlocal;
    L.n_if_start_pad = start;
    L.n_shared = start + nif;
  }

  L.n_int_pad = (int_cursor + 15) & ~(int64_t)15;
  L.npairs = L.blk_sh_off[L.nblocks];
  L.n_internal = (L.n_int_pad + L.n_shared + 15) & ~(int64_t)15;
  if (L.n_internal > 2000000000ll)
    return "local vector exceeds int32 indexing";
  for (int64_t g = 0; g < ndofs; ++g)
    if (sh_id[g] >= 0)
      L.dof_perm[g] = (int32_t)(L.n_int_pad + sh_id[g]);
  for (int64_t k = 0; k < L.npairs; ++k)
    L.sh_gidx[k] = (int32_t)(L.n_int_pad + pair_shid[k]);

  // shared dof -> pairs CSR, ascending pair (= block) order
  L.sh_ptr.assign(L.n_shared + 1, 0);
  for (int64_t k = 0; k < L.npairs; ++k)
    ++L.sh_ptr[pair_shid[k] + 1];
  for (int64_t s = 0; s < L.n_shared; ++s)
    L.sh_ptr[s + 1] += L.sh_ptr[s];
  L.sh_pairs.resize(L.npairs);
  {
    std::vector<int64_t> cur(L.sh_ptr.begin(), L.sh_ptr.end() - 1);
    for (int64_t k = 0; k < L.npairs; ++k)
      L.sh_pairs[cur[pair_shid[k]]++] = k;
  }
  // Where a pair's partial sum lives (pair_pos).  Rank-local shared dof s, its j-th sharing block (ascending
  // block order): plane_off[j] + s -- plane j holds one value for every dof with more than j sharers, a prefix of
  // the local range, so the reduction reads plane after plane at the dof's own index, coalesced and without an
  // index list.  Pairs of interface dofs follow the planes in CSR order.
  L.plane_cnt.clear(), L.plane_off.clear();
  int64_t cursor = 0;
  for (int j = 0;; ++j)
  {
    int64_t cnt = 0;  // dofs with more than j sharers: a prefix (descending order), found by bisection
    {
      int64_t lo = 0, hi = L.n_shared_local;
      while (lo < hi)
      {
        const int64_t mid = (lo + hi) / 2;
        if (L.sh_ptr[mid + 1] - L.sh_ptr[mid] > j)
          lo = mid + 1;
        else
          hi = mid;
      }
      cnt = lo;
    }
    if (cnt == 0)
      break;
    L.plane_cnt.push_back(cnt), L.plane_off.push_back(cursor);
    cursor = (cursor + cnt + 15) & ~(int64_t)15;
  }
  L.pair_pos.assign(L.npairs, -1);
  for (int64_t sidx = 0; sidx < L.n_shared_local; ++sidx)
    for (int64_t k = L.sh_ptr[sidx]; k < L.sh_ptr[sidx + 1]; ++k)
      L.pair_pos[L.sh_pairs[k]] = (int32_t)(L.plane_off[k - L.sh_ptr[sidx]] + sidx);
  const int64_t if_first = L.sh_ptr[std::min(L.n_if_start_pad, L.n_shared)];
  for (int64_t k = if_first; k < L.npairs; ++k)
    L.pair_pos[L.sh_pairs[k]] = (int32_t)(cursor + (k - if_first));
  L.n_partial = cursor + (L.npairs - if_first);
  if (L.n_partial + L.n_shared > 2000000000ll)
    return "partial slab exceeds int32 indexing";
  return "";
}

std::string verify_layout(const Layout& L, const int32_t* dm)
{
  const int Nd = L.Nd;
  // cell_perm is a permutation
  {
    std::vector<uint8_t> seen(L.ncells, 0);
    for (int64_t e = 0; e < L.ncells; ++e)
    {
      const int32_t c = L.cell_perm[e];
      if (c < 0 || c >= L.ncells || seen[c])
        return "cell_perm is not a permutation";
      seen[c] = 1;
    }
  }
  // dof_perm is injective into [0, n_internal)
  {
    std::vector<uint8_t> seen(L.n_internal, 0);
    for (int64_t g = 0; g < L.ndofs; ++g)
    {
      const int32_t p = L.dof_perm[g];
      if (p < 0 || p >= L.n_internal || seen[p])
        return "dof_perm is not injective";
      seen[p] = 1;
    }
  }
  int64_t pairs_seen = 0;
  for (int32_t b = 0; b < L.nblocks; ++b)
  {
    const Layout::Shape& sh = L.shapes[L.blk_shape[b]];
    if (sh.nelem != L.blk_elem_off[b + 1] - L.blk_elem_off[b])
      return "shape/element count mismatch";
    if (sh.nloc - sh.nint != L.blk_sh_off[b + 1] - L.blk_sh_off[b])
      return "shape/shared count mismatch";
    if (L.blk_int_off[b] % 16)
      return "interior range not 128-byte aligned";
    // local dofmap maps back to the caller dofmap through the internal numbering
    for (int32_t e = 0; e < sh.nelem; ++e)
    {
      const int32_t* d = dm + (int64_t)L.cell_perm[L.blk_elem_off[b] + e] * Nd;
      for (int i = 0; i < Nd; ++i)
      {
        const int32_t l = L.ldm[sh.ldm_off + (int64_t)e * Nd + i];
        if (l >= sh.nloc)
          return "local index out of range";
        const int32_t internal = l < sh.nint ? L.blk_int_off[b] + l
                                             : L.sh_gidx[L.blk_sh_off[b] + (l - sh.nint)];
        if (internal != L.dof_perm[d[i]])
          return "local dofmap inconsistent with dof_perm";
      }
    }
    // rounds: every element exactly once, no dof shared inside a round
    std::vector<int> used(sh.nelem, 0);
    std::vector<int32_t> stamp(sh.nloc, -1);
    for (int32_t r = 0; r < sh.nrounds; ++r)
      for (int s = 0; s < L.slots; ++s)
      {
        const int e = L.rounds[sh.rounds_off + (int64_t)r * L.slots + s];
        if (e < 0)
          continue;
        if (e >= sh.nelem || used[e]++)
          return "round table does not cover each element once";
        for (int i = 0; i < Nd; ++i)
          if (stamp[L.ldm[sh.ldm_off + (int64_t)e * Nd + i]] == r)
            return "two elements of one round share a dof";
        for (int i = 0; i < Nd; ++i)
          stamp[L.ldm[sh.ldm_off + (int64_t)e * Nd + i]] = r;
      }
    for (int e = 0; e < sh.nelem; ++e)
      if (!used[e])
        return "element missing from rounds";
    pairs_seen += sh.nloc - sh.nint;
  }
  if (pairs_seen != L.npairs)
    return "pair count mismatch";
  {
    std::vector<char> taken(L.n_partial, 0);
    for (int64_t k = 0; k < L.npairs; ++k)
    {
      if (L.pair_pos[k] < 0 || L.pair_pos[k] >= L.n_partial || taken[L.pair_pos[k]]++)
        return "two pairs share a partial position";
    }
  }
  for (int64_t s = 0; s < L.n_shared; ++s)
    for (int64_t k = L.sh_ptr[s]; k < L.sh_ptr[s + 1]; ++k)
    {
      if (L.sh_gidx[L.sh_pairs[k]] != L.n_int_pad + s)
        return "shared CSR inconsistent";
      if (k > L.sh_ptr[s] && L.sh_pairs[k] <= L.sh_pairs[k - 1])
        return "shared CSR not in ascending block order";
      const int64_t pos = L.pair_pos[L.sh_pairs[k]], j = k - L.sh_ptr[s];
      if (pos < 0 || pos >= L.n_partial)
        return "partial position out of range";
      if (s < L.n_shared_local && (j >= (int64_t)L.plane_cnt.size() || s >= L.plane_cnt[j] || pos != L.plane_off[j] + s))
        return "partial position is not (plane of the sharer's rank, dof index)";
    }
  return "";
}

} // namespace fus
