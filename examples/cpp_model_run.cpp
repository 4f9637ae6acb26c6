// cpp_model_run.cpp -- drives the C++ host classes of include/fusmi.hpp (StiffnessSpectral3D,
// MassSpectral3D, Linear/Lossy/WesterveltSpectral3D) the way the reference's mains drive theirs
// (cpp/fenicsx-sf/benchmarks/PH1/BM7-SC1/main.cpp:121-130: construct, init(), rk4(), u_sol()).
// Mesh and coefficients come from a flat binary file (written by tests/test_cpp_host.py); results
// go to another.  Usage: cpp_model_run <in.bin> <out.bin>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fusmi.hpp"

namespace
{
struct Reader
{
  FILE* f;
  template <typename U>
  std::vector<U> arr(size_t n)
  {
    std::vector<U> v(n);
    if (n && fread(v.data(), sizeof(U), n, f) != n)
    {
      fprintf(stderr, "short read\n");
      exit(2);
    }
    return v;
  }
};

template <int P>
int run(Reader& r, const std::vector<int64_t>& h, const std::vector<double>& s, const char* outpath)
{
  using T = double;
  const int tdim = (int)h[0];
  const int64_t ncells = h[2], ndofs = h[3], nnodes = h[4], nfacets = h[5], kind = h[6], nsteps = h[7];
  const int N = P + 1, Nd = tdim == 3 ? N * N * N : N * N, nv = tdim == 3 ? 8 : 4;
  auto tdm = r.arr<int32_t>((size_t)ncells * Nd);
  auto nodes = r.arr<double>(N);
  auto gx = r.arr<double>((size_t)nnodes * 3);
  auto gdm = r.arr<int32_t>((size_t)ncells * nv);
  auto fc = r.arr<int32_t>(nfacets), fl = r.arr<int32_t>(nfacets), ft = r.arr<int32_t>(nfacets);
  auto c0 = r.arr<double>(ncells), rho0 = r.arr<double>(ncells), delta0 = r.arr<double>(ncells),
       beta0 = r.arr<double>(ncells);
  auto x = r.arr<double>(ndofs), coeffs = r.arr<double>(ncells);

  auto ctx = std::make_shared<fusmi::Context>(0);
  fusmi::SpaceView<T> V;
  V.tdim = tdim, V.ncells = ncells, V.ndofs = ndofs, V.nnodes = nnodes;
  V.tensor_dofmap = tdm.data(), V.nodes1d = nodes.data(), V.geom_x = gx.data(), V.geom_dofmap = gdm.data();
  auto data = std::make_shared<fusmi::SpectralOperatorData<T, P>>(ctx, V, kind == 0 ? 1 : 2);

  // operators: y += A(coeffs) x, y starts at 1 to show the accumulation
  std::vector<T> ys(ndofs, 1.0), ym(ndofs, 1.0);
  fusmi::StiffnessSpectral3D<T, P> stiffness(data);
  fusmi::MassSpectral3D<T, P> mass(data);
  stiffness(x.data(), coeffs.data(), ys.data());
  mass(x.data(), coeffs.data(), ym.data());

  fusmi::FacetView facets{nfacets, fc.data(), fl.data(), ft.data()};
  const T freq = s[0], amp = s[1], speed = s[2], dt = s[3];
  std::vector<T> u, v;
  int64_t taken = 0, nd = 0;
  const T tf = dt * nsteps * (1.0 - 1e-9);
  auto solve = [&](auto& model)
  {
    model.init();
    taken = model.rk4(0.0, tf, dt);
    u = model.u_sol(), v = model.v_sol(), nd = model.number_of_dofs();
  };
  if (kind == 0)
  {
    fusmi::LinearSpectral3D<T, P> model(data, facets, c0.data(), rho0.data(), freq, amp, speed);
    solve(model);
  }
  else if (kind == 1)
  {
    fusmi::LossySpectral3D<T, P> model(data, facets, c0.data(), rho0.data(), delta0.data(), freq, amp, speed);
    solve(model);
  }
  else
  {
    fusmi::WesterveltSpectral3D<T, P> model(data, facets, c0.data(), rho0.data(), delta0.data(), beta0.data(),
                                            freq, amp, speed);
    solve(model);
  }
  FILE* o = fopen(outpath, "wb");
  const int64_t tail[2] = {taken, nd};
  fwrite(ys.data(), 8, ndofs, o), fwrite(ym.data(), 8, ndofs, o), fwrite(u.data(), 8, ndofs, o),
      fwrite(v.data(), 8, ndofs, o), fwrite(tail, 8, 2, o);
  fclose(o);
  // what the examples' mains print around the run: global smallest cell size (mesh::h + MPI_MIN,
  // linear_planewave2d_1/main.cpp:60-68) and the L2 norm of the solution (:151-157)
  const double hmin = ctx->allreduce(data->hmin(), FUS_MIN);
  const double l2 = std::sqrt(ctx->allreduce(data->norm2(u.data()), FUS_SUM));
  printf("hmin %.17g L2 norm %.17g\n", hmin, l2);
  printf("ok: %lld steps, %lld dofs\n", (long long)taken, (long long)nd);
  return 0;
}
} // namespace

int main(int argc, char** argv)
{
  if (argc != 3)
  {
    fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]);
    return 2;
  }
  Reader r{fopen(argv[1], "rb")};
  if (!r.f)
    return 2;
  auto h = r.arr<int64_t>(8);   // tdim, P, ncells, ndofs, nnodes, nfacets, kind, nsteps
  auto s = r.arr<double>(4);    // freq, amp, speed, dt
  try
  {
    switch (h[1])
    {
    case 2: return run<2>(r, h, s, argv[2]);
    case 3: return run<3>(r, h, s, argv[2]);
    case 4: return run<4>(r, h, s, argv[2]);
    case 5: return run<5>(r, h, s, argv[2]);
    case 6: return run<6>(r, h, s, argv[2]);
    case 7: return run<7>(r, h, s, argv[2]);
    default: fprintf(stderr, "unsupported degree\n"); return 2;
    }
  }
  catch (const fusmi::Error& e)
  {
    fprintf(stderr, "fusmi error %d: %s\n", e.code, e.what());
    return e.code == FUS_ERR_HIP ? 3 : 1;   // 3: no device (expected on a CPU-only host)
  }
}
