// layout_driver.cpp -- runs the host-side block-layout builder (csrc/layout.cpp) on a dofmap read
// from a binary file; built by tests/test_layout_sanitized.py with -fsanitize=address,undefined.
// File: int64 {tdim, P, ncells, ndofs, block_elems, waves, has_mask}; int32 dofmap[ncells*Nd];
// double centroids[ncells*3]; uint8 mask[ndofs] if has_mask.
#include <cstdint>
#include <cstdio>
#include <vector>

#include "layout.hpp"

int main(int argc, char** argv)
{
  if (argc != 2)
    return 2;
  FILE* f = fopen(argv[1], "rb");
  if (!f)
    return 2;
  int64_t h[7];
  if (fread(h, 8, 7, f) != 7)
    return 2;
  const int tdim = (int)h[0], P = (int)h[1], N = P + 1;
  const int64_t nc = h[2], nd = h[3];
  const int Nd = tdim == 3 ? N * N * N : N * N;
  std::vector<int32_t> dm((size_t)nc * Nd);
  std::vector<double> cen((size_t)nc * 3);
  std::vector<uint8_t> mask(h[6] ? (size_t)nd : 0);
  if (fread(dm.data(), 4, dm.size(), f) != dm.size() || fread(cen.data(), 8, cen.size(), f) != cen.size()
      || (h[6] && fread(mask.data(), 1, mask.size(), f) != mask.size()))
    return 2;
  fclose(f);
  fus::Layout L;
  std::string err = fus::build_layout(L, P, nc, nd, dm.data(), cen.data(), (int)h[4], (int)h[5],
                                      h[6] ? mask.data() : nullptr, tdim);
  if (!err.empty())
  {
    printf("build: %s\n", err.c_str());
    return 1;
  }
  err = fus::verify_layout(L, dm.data());
  if (!err.empty())
  {
    printf("verify: %s\n", err.c_str());
    return 1;
  }
  printf("ok blocks=%d if=%d interior=%lld shared=%lld pairs=%lld lds=%zu\n", L.nblocks, L.nblocks_if,
         (long long)L.n_interior, (long long)L.n_shared, (long long)L.npairs, L.lds_bytes(8));
  return 0;
}
