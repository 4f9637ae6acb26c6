// layout.hpp -- host-side block layout of a spectral-element mesh for the LDS-blocked operator.
//
// The reference walks cells serially and scatter-adds into the global vector
// (cpp/fenicsx-sf/common/spectral_op.hpp:183-242).  On MI355X the cells are grouped into
// LDS-sized blocks: a workgroup gathers the block's DOFs once, accumulates every element's
// contribution in LDS, and writes each DOF once.  DOFs touched by a single block ("interior") are
// complete when the block finishes; DOFs touched by several blocks ("shared") are written as
// per-(block, dof) partial sums and reduced in a fixed order by a second kernel -- no atomics,
// bitwise reproducible.  Vectors live in an INTERNAL numbering: each block's interior DOFs are
// contiguous (128-byte aligned), shared DOFs follow.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace fus
{

struct Layout
{
  int P = 0, N = 0, Nd = 0;  // Nd = N^tdim nodes per element
  int tdim = 3;
  int waves = 4;   // waves per workgroup
  int epw = 1;     // elements per wave = 64 / N^2 (at least 1)
  int slots = 4;   // elements per round = waves * epw
  int64_t ncells = 0, ndofs = 0;

  // --- cells ---
  int32_t nblocks = 0;
  int32_t nblocks_if = 0;             // blocks [0, nblocks_if) touch interface dofs (held by other ranks too)
  std::vector<int32_t> cell_perm;     // [ncells] internal element -> caller cell
  std::vector<int32_t> blk_elem_off;  // [nblocks+1] first internal element of block

  // --- block shapes (deduplicated: structured meshes have a handful) ---
  struct Shape
  {
    int32_t nelem, nloc, nint, nrounds;
    int64_t rounds_off;  // into `rounds`  (nrounds*slots entries, block-relative element or -1)
    int64_t ldm_off;     // into `ldm`     (nelem*Nd entries, [elem][a][p] local dof index)
  };
  std::vector<Shape> shapes;
  std::vector<int32_t> blk_shape;     // [nblocks]
  std::vector<int16_t> rounds;        // block-relative element index per (round, slot), -1 = idle
  std::vector<uint16_t> ldm;          // local dofmaps

  // --- DOF numbering ---
  std::vector<int32_t> dof_perm;      // [ndofs] caller dof -> internal dof
  std::vector<int32_t> blk_int_off;   // [nblocks] internal index of the block's first interior dof
  std::vector<int64_t> blk_sh_off;    // [nblocks+1] offset of the block's shared slots (pairs)
  std::vector<int32_t> sh_gidx;       // [npairs] internal dof of each (block, shared slot)
  int64_t n_int_pad = 0;              // shared region starts here (multiple of 16)
  int64_t n_interior = 0, n_shared = 0, npairs = 0;
  // shared slots [0, n_shared_local) are rank-local dofs; [n_if_start_pad, n_shared) are interface
  // dofs (held by other ranks too); n_if_start_pad is a multiple of 16, slots in between are empty.
  // n_shared counts slots (= dofs when there is no interface).
  int64_t n_shared_local = 0, n_if_start_pad = 0;
  int64_t n_internal = 0;             // padded internal vector length (multiple of 16)
  std::vector<int64_t> sh_ptr;        // [n_shared+1] CSR: shared dof -> pair indices
  std::vector<int64_t> sh_pairs;      // [npairs] ascending block order
  // position of each pair's partial sum in the slab of n_partial values: plane_off[j] + s for the j-th sharer
  // of rank-local shared dof s (plane j covers the plane_cnt[j] dofs with more than j sharers: a prefix of the
  // local range, which is ordered by descending sharer count); pairs of interface dofs follow the planes
  std::vector<int32_t> pair_pos;      // [npairs]
  std::vector<int64_t> plane_cnt, plane_off;
  int64_t n_partial = 0;
  int32_t max_nloc = 0, max_rounds = 0, max_nelem = 0;

  // nfields: operator inputs staged per block (1: Linear; 2: Lossy, K(c1) u + K(c2) v in one pass)
  size_t lds_bytes(size_t sizeofT, int nfields = 1, int geom_cell_stride = 0) const
  {
    const size_t ne = ((size_t)max_nelem + 7) & ~(size_t)7;
    return (size_t)nfields * ((max_nloc + 1) & ~1) * sizeofT          // x_l (, x2_l)
           + (size_t)((max_nloc + 1) & ~1) * 8                        // y_l: fp64 for every T (kernels.hpp FUS_ACC)
           + (size_t)slots * ((tdim == 3 && N == 7 && sizeofT == 4) ? N * 56 : Nd + N) * sizeofT   // per-element exchange tile (kernels.hpp tile_slot_entries)
           + (size_t)N * N * sizeofT + nfields * ne * sizeofT  // derivative table, coefficients
           + (geom_cell_stride ? ((size_t)geom_cell_stride * ne + (N <= 8 ? 16 : 24)) * sizeofT : 0)  // per-cell geometry, 1-D weights + points (8 or 12 slots each)
           + ne * Nd * 2                                 // local dofmaps
           + (size_t)max_rounds * slots * 2 + 16;        // round table
  }
};

// Builds the layout.  centroids: [ncells*3] (any consistent coordinates, z = 0 for quadrilaterals; used by the recursive
// coordinate bisection that forms compact blocks).  force_shared (optional, [ndofs]): dofs that must
// be classified shared even if a single local block touches them (dofs held by other ranks too).
// Returns empty string or an error message.
std::string build_layout(Layout& L, int P, int64_t ncells, int64_t ndofs,
                         const int32_t* tensor_dofmap, const double* centroids, int block_elems,
                         int waves, const uint8_t* force_shared = nullptr, int tdim = 3, int slot_factor = 1);

// Internal consistency check used by fus_layout_check and the CPU tests.
std::string verify_layout(const Layout& L, const int32_t* tensor_dofmap);

} // namespace fus
