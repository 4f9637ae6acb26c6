// fusmi.hpp -- C++ host side above the C ABI (fusmi.h): the reference's operator and model classes
// with the same names, template parameters and call semantics, minus the DOLFINx types.
//
//   reference (cpp/fenicsx-sf/common)                     here (namespace fusmi)
//   StiffnessSpectral3D<T,P>(V); op(x, coeffs, y)         StiffnessSpectral3D<T,P>(data); op(x, coeffs, y)
//     spectral_op.hpp:132-243  (y += K(coeffs) x)
//   MassSpectral3D<T,P>(V); op(x, coeffs, y)              MassSpectral3D<T,P>(data); op(x, coeffs, y)
//     spectral_op.hpp:29-86
//   StiffnessSpectral2D / MassSpectral2D                  same classes, SpaceView::tdim = 2
//     cpp/fenicsx-sf-naive/common/spectral_op.hpp:29-359
//   LinearSpectral3D<T,P>(element, mesh, facet_tags,      LinearSpectral3D<T,P>(data, facets, c0, rho0,
//       c0, rho0, freq, amp, speed)                            freq, amp, speed)
//     init(); rk4(t0, tf, dt); u_sol(); number_of_dofs()    init(); rk4(t0, tf, dt); u_sol(); number_of_dofs()
//     Linear.hpp:52-347
//   LossySpectral3D (Lossy.hpp:56-342), WesterveltSpectral3D (Westervelt.hpp:58-373): + delta0 (, beta0)
//
// Where the reference takes a dolfinx::fem::FunctionSpace / Mesh / MeshTags and derives arrays from
// them (spectral_op.hpp:135-171, Linear.hpp:113-118), these classes take those arrays directly
// (SpaceView, FacetView): INTEGRATION.md shows how a DOLFINx build fills them.  Vectors are plain
// pointers to caller-owned host memory (la::Vector::array() in the reference).  Errors become
// fusmi::Error (the C ABI's code and message); nothing else is added on top of the C ABI.
// Header-only, C++17, no dependency beyond fusmi.h.
#pragma once
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

#include "fusmi.h"

namespace fusmi
{

struct Error : std::runtime_error
{
  int code;
  Error(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};

inline void check(int rc)
{
  if (rc != FUS_OK)
    throw Error(rc, fus_last_error());
}

template <typename T>
constexpr int dtype_code()
{
  static_assert(std::is_same<T, double>::value || std::is_same<T, float>::value, "T is float or double");
  return std::is_same<T, double>::value ? FUS_F64 : FUS_F32;
}

// One GPU = one context (the reference: one MPI rank).
class Context
{
public:
  explicit Context(int device = 0) { check(fus_init(device, &h_)); }
  ~Context() { fus_finalize(h_); }
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  void set_option(const char* key, std::int64_t value) { check(fus_set_option(h_, key, value)); }
  void synchronize() { check(fus_synchronize(h_)); }
  // multi-GPU: id from comm_unique_id() on rank 0, broadcast by the caller (MPI_Bcast in the reference's setting)
  static std::vector<char> comm_unique_id()
  {
    std::vector<char> id(128);
    check(fus_comm_unique_id(id.data()));
    return id;
  }
  void comm_init(int rank, int nranks, const std::vector<char>& id) { check(fus_comm_init(h_, rank, nranks, id.data())); }
  // MPI_Reduce + MPI_Bcast of the examples' mains (linear_planewave2d_1/main.cpp:67-68): op = FUS_SUM | FUS_MIN | FUS_MAX
  double allreduce(double v, int op)
  {
    check(fus_comm_allreduce(h_, &v, 1, op));
    return v;
  }
  fus_ctx* handle() const { return h_; }

private:
  fus_ctx* h_ = nullptr;
};

// What the reference's operator constructor derives from the function space (spectral_op.hpp:135-171):
// tensor-ordered cell dofmap (reorder_dofmap, permute.hpp:15-42), 1-D GLL node coordinates in the
// element's local order, mesh geometry.
template <typename T>
struct SpaceView
{
  int tdim = 3;                        // 3: hexahedra, 2: quadrilaterals
  std::int64_t ncells = 0, ndofs = 0, nnodes = 0;
  const std::int32_t* tensor_dofmap = nullptr;  // [ncells * (P+1)^tdim]
  const double* nodes1d = nullptr;              // [P+1]
  const T* geom_x = nullptr;                    // [nnodes * 3]
  const std::int32_t* geom_dofmap = nullptr;    // [ncells * 2^tdim] (order 1) or [ncells * 27] (order 2, tensor order)
  int geom_order = 1;
  // shared dofs per neighbour rank (IndexMap data behind scatter_fwd/rev, Linear.hpp:196-206); empty on one rank
  std::vector<std::int32_t> neighbour_ranks;
  std::vector<std::int64_t> neighbour_counts;
  std::vector<std::int32_t> neighbour_dofs;
};

// Boundary facets as (cell, local facet) pairs with their tag: what fem::compute_integration_domains
// returns per tag (Linear.hpp:113-118); tag 1 = source, tag 2 = absorbing (forms.py:36-39).
struct FacetView
{
  std::int64_t nfacets = 0;
  const std::int32_t* cells = nullptr;
  const std::int32_t* local_facets = nullptr;
  const std::int32_t* tags = nullptr;
};

// Device-resident operator data shared by the mass and stiffness operators and the models (the
// reference builds one copy per operator object, Lossy.hpp:152-153).
template <typename T, int P>
class SpectralOperatorData
{
public:
  // fields = 2 for the lossy / Westervelt models (two operator inputs per block pass)
  SpectralOperatorData(std::shared_ptr<Context> ctx, const SpaceView<T>& V, int fields = 1) : ctx_(std::move(ctx))
  {
    ctx_->set_option("fields", fields);
    int rc = fus_op_create(ctx_->handle(), V.tdim, P, dtype_code<T>(), V.ncells, V.ndofs, V.tensor_dofmap,
                           V.nodes1d, V.geom_x, V.nnodes, V.geom_dofmap, V.geom_order, &h_);
    fus_set_option(ctx_->handle(), "fields", 1);
    check(rc);
    ncells_ = V.ncells, ndofs_ = V.ndofs, tdim_ = V.tdim;
    if (!V.neighbour_ranks.empty())
      check(fus_op_set_neighbours(h_, (int)V.neighbour_ranks.size(), V.neighbour_ranks.data(),
                                  V.neighbour_counts.data(), V.neighbour_dofs.data()));
  }
  ~SpectralOperatorData() { fus_op_destroy(h_); }
  SpectralOperatorData(const SpectralOperatorData&) = delete;
  SpectralOperatorData& operator=(const SpectralOperatorData&) = delete;
  fus_op* handle() const { return h_; }
  const std::shared_ptr<Context>& context() const { return ctx_; }
  std::int64_t ncells() const { return ncells_; }
  std::int64_t ndofs() const { return ndofs_; }
  int tdim() const { return tdim_; }
  bool is_affine() const { return fus_op_is_affine(h_) != 0; }
  int geometry_mode() const { return fus_op_geometry_mode(h_); }  // 0 streamed, 1 affine, 2 trilinear
  // smallest local cell size, mesh::h of linear_planewave2d_1/main.cpp:60-64 (global: Context::allreduce(h, FUS_MIN))
  double hmin() const
  {
    double h = 0;
    check(fus_op_hmin(h_, &h));
    return h;
  }
  // local part of the squared L2 norm (assemble_scalar(u*u*dx), main.cpp:151-157); sum over ranks
  double norm2(const T* x, int loc = FUS_HOST) const
  {
    double v = 0;
    check(fus_op_norm2(h_, x, loc, &v));
    return v;
  }

private:
  std::shared_ptr<Context> ctx_;
  fus_op* h_ = nullptr;
  std::int64_t ncells_ = 0, ndofs_ = 0;
  int tdim_ = 3;
};

// y += K(coeffs) x ; x [ndofs], coeffs [ncells] (one scalar per cell), y [ndofs] accumulated
// (spectral_op.hpp:173-243; the caller zeroes y, Linear.hpp:203).
template <typename T, int P>
class StiffnessSpectral3D
{
public:
  explicit StiffnessSpectral3D(std::shared_ptr<SpectralOperatorData<T, P>> data) : d_(std::move(data)) {}
  void operator()(const T* x, const T* coeffs, T* y) const
  {
    check(fus_stiffness_apply(d_->handle(), x, coeffs, y, FUS_HOST));
  }

private:
  std::shared_ptr<SpectralOperatorData<T, P>> d_;
};

// y += M(coeffs) x (spectral_op.hpp:69-86)
template <typename T, int P>
class MassSpectral3D
{
public:
  explicit MassSpectral3D(std::shared_ptr<SpectralOperatorData<T, P>> data) : d_(std::move(data)) {}
  void operator()(const T* x, const T* coeffs, T* y) const
  {
    check(fus_mass_apply(d_->handle(), x, coeffs, y, FUS_HOST));
  }

private:
  std::shared_ptr<SpectralOperatorData<T, P>> d_;
};

template <typename T, int P>
using StiffnessSpectral2D = StiffnessSpectral3D<T, P>;  // SpaceView::tdim = 2
template <typename T, int P>
using MassSpectral2D = MassSpectral3D<T, P>;

namespace detail
{
template <typename T, int P, int KIND>
class SpectralModel
{
public:
  void init() { check(fus_model_init(h_)); }  // Linear.hpp:161-164: u = v = 0
  // classical RK4 from t0 until t >= tf (Linear.hpp:228-314); returns the number of steps taken
  std::int64_t rk4(const T& t0, const T& tf, const T& dt)
  {
    std::int64_t nsteps = 0;
    check(fus_model_rk4(h_, (double)t0, (double)tf, (double)dt, &nsteps));
    return nsteps;
  }
  // exactly nsteps steps (no last shortened step)
  void rk4_steps(const T& t0, const T& dt, std::int64_t nsteps) { check(fus_model_rk4_steps(h_, (double)t0, (double)dt, nsteps)); }
  // explicit RK order 1..4 of the Python reference (python/src/fenicsxfus/_linear.py:286-311)
  void set_rk_order(int order) { check(fus_model_set_rk_order(h_, order)); }
  std::vector<T> u_sol() const { return get(FUS_U); }  // Linear.hpp:316
  std::vector<T> v_sol() const { return get(FUS_V); }
  void set_state(const T* u, const T* v)
  {
    if (u)
      check(fus_model_set(h_, FUS_U, u, FUS_HOST));
    if (v)
      check(fus_model_set(h_, FUS_V, v, FUS_HOST));
  }
  std::vector<T> mass_vector() const
  {
    std::vector<T> m((size_t)d_->ndofs());
    check(fus_model_get_mass(h_, m.data()));
    return m;
  }
  std::int64_t number_of_dofs() const { return fus_model_ndofs(h_); }  // Linear.hpp:318
  // receivers: u->eval(points_on_proc, shape, cells, u_eval, ...) of cpp/mwe/parallel_eval_line/main.cpp:49-84 on the
  // resident solution -- cells / reference coordinates located by the caller (points outside the rank dropped)
  void set_receivers(const std::vector<std::int32_t>& cells, const std::vector<double>& refcoords)
  {
    nrecv_ = (std::int64_t)cells.size();
    check(fus_model_set_receivers(h_, nrecv_, cells.data(), refcoords.data()));
  }
  std::vector<T> eval(int which = FUS_U) const
  {
    std::vector<T> out((size_t)nrecv_);
    check(fus_model_sample(h_, which, out.data(), FUS_HOST));
    return out;
  }
  // sample after every `every`-th step of rk4 / rk4_steps into a device buffer of `capacity` records
  void record(int every, std::int64_t capacity, int which = FUS_U) { check(fus_model_record(h_, which, every, capacity)); }
  std::vector<T> records(std::vector<double>* times = nullptr) const
  {
    std::int64_t n = 0;
    check(fus_model_get_records(h_, nullptr, nullptr, &n));
    std::vector<T> out((size_t)(n * nrecv_));
    std::vector<double> t((size_t)n);
    check(fus_model_get_records(h_, out.data(), t.data(), &n));
    if (times)
      *times = t;
    return out;
  }
  fus_model* handle() const { return h_; }
  ~SpectralModel() { fus_model_destroy(h_); }
  SpectralModel(const SpectralModel&) = delete;
  SpectralModel& operator=(const SpectralModel&) = delete;

protected:
  SpectralModel(std::shared_ptr<SpectralOperatorData<T, P>> data, const FacetView& f, const T* c0, const T* rho0,
                const T* delta0, const T* beta0, const T& freq, const T& amp, const T& speed)
      : d_(std::move(data))
  {
    check(fus_model_create(d_->context()->handle(), KIND, d_->handle(), c0, rho0, delta0, beta0, f.nfacets, f.cells,
                           f.local_facets, f.tags, (double)freq, (double)amp, (double)speed, &h_));
  }

private:
  std::vector<T> get(int which) const
  {
    std::vector<T> out((size_t)d_->ndofs());
    check(fus_model_get(h_, which, out.data(), FUS_HOST));
    return out;
  }
  std::shared_ptr<SpectralOperatorData<T, P>> d_;
  fus_model* h_ = nullptr;
  std::int64_t nrecv_ = 0;
};
} // namespace detail

// c0, rho0 (, delta0, beta0): one value per cell (the reference's DG0 functions); freq, amp, speed:
// sourceFrequency, sourceAmplitude, sourceSpeed of Linear.hpp:55-62.
template <typename T, int P>
class LinearSpectral3D : public detail::SpectralModel<T, P, FUS_LINEAR>
{
public:
  LinearSpectral3D(std::shared_ptr<SpectralOperatorData<T, P>> data, const FacetView& facets, const T* c0,
                   const T* rho0, const T& freq, const T& amp, const T& speed)
      : detail::SpectralModel<T, P, FUS_LINEAR>(std::move(data), facets, c0, rho0, nullptr, nullptr, freq, amp, speed)
  {
  }
};
template <typename T, int P>
using LinearSpectral2D = LinearSpectral3D<T, P>;  // cpp/fenicsx-sf-naive/common/Linear.hpp:52-350

// data must have been created with fields = 2.  Boundary forms: Context::set_option("forms", 0 | 1)
// before construction (fusmi.h).
template <typename T, int P>
class LossySpectral3D : public detail::SpectralModel<T, P, FUS_LOSSY>
{
public:
  LossySpectral3D(std::shared_ptr<SpectralOperatorData<T, P>> data, const FacetView& facets, const T* c0,
                  const T* rho0, const T* delta0, const T& freq, const T& amp, const T& speed)
      : detail::SpectralModel<T, P, FUS_LOSSY>(std::move(data), facets, c0, rho0, delta0, nullptr, freq, amp, speed)
  {
  }
};

template <typename T, int P>
class WesterveltSpectral3D : public detail::SpectralModel<T, P, FUS_WESTERVELT>
{
public:
  WesterveltSpectral3D(std::shared_ptr<SpectralOperatorData<T, P>> data, const FacetView& facets, const T* c0,
                       const T* rho0, const T* delta0, const T* beta0, const T& freq, const T& amp,
                       const T& speed)
      : detail::SpectralModel<T, P, FUS_WESTERVELT>(std::move(data), facets, c0, rho0, delta0, beta0, freq, amp,
                                                    speed)
  {
  }
};

// delta = 2 alpha c0^3 / w0^2, alpha in Np/m (Lossy.hpp:375-379)
template <typename T>
T compute_diffusivity_of_sound(const T w0, const T c0, const T alpha)
{
  return 2 * alpha * c0 * c0 * c0 / w0 / w0;
}

} // namespace fusmi
