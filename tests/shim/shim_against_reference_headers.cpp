// Compile-and-link check (build container only): include/ftkx_shim.hh instantiated with the REAL ftk::lattice and
// ftk::feature_point_lite_t from the reference tree, behind functions that have exactly the signatures the reference declares
// for its accelerator back-ends (critical_point_tracker_2d_regular.hh:33-63, critical_point_tracker_3d_regular.hh:42-56) plus the
// scaling factor.  Nothing is executed here (no GPU in the build container).
#include <cmath>
#include <ftk/mesh/lattice.hh>
#include <ftk/features/feature_point_lite.hh>
#include <ftkx_shim.hh>

std::vector<ftk::feature_point_lite_t> extract_cp2dt_hip(
    int scope, int current_timestep, const ftk::lattice &domain, const ftk::lattice &core, const ftk::lattice &ext,
    const double *Vc, const double *Vn, const double *Jc, const double *Jn, const double *Sc, const double *Sn,
    bool use_explicit_coords, const double *coords, unsigned long long factor, bool symmetric, bool use_type_filter, unsigned type_filter)
{
  return ftkx::extract_cp2dt_hip<ftk::feature_point_lite_t>(scope, current_timestep, domain, core, ext, Vc, Vn, Jc, Jn, Sc, Sn,
                                                            use_explicit_coords, coords, factor, symmetric, use_type_filter, type_filter);
}

std::vector<ftk::feature_point_lite_t> extract_cp3dt_hip(
    int scope, int current_timestep, const ftk::lattice &domain4, const ftk::lattice &core4, const ftk::lattice &ext3,
    const double *Vc, const double *Vl, const double *Jc, const double *Jl, const double *Sc, const double *Sl,
    unsigned long long factor, bool symmetric, bool robust)
{
  return ftkx::extract_cp3dt_hip<ftk::feature_point_lite_t>(scope, current_timestep, domain4, core4, ext3, Vc, Vl, Jc, Jl, Sc, Sl, factor, symmetric, robust);
}

int main(int argc, char **)
{
  static_assert(sizeof(ftk::feature_point_lite_t) == 72, "feature_point_lite_t");
  static_assert(offsetof(ftk::feature_point_lite_t, tag) == offsetof(ftkx_cp_t, tag) && offsetof(ftk::feature_point_lite_t, type) == offsetof(ftkx_cp_t, type)
                && offsetof(ftk::feature_point_lite_t, scalar) == offsetof(ftkx_cp_t, scalar) && offsetof(ftk::feature_point_lite_t, t) == offsetof(ftkx_cp_t, t), "field offsets");
  if (argc > 100) {   // referenced, never run
    ftk::lattice d3({2, 2, 0}, {5, 5, 1}), e2({0, 0}, {8, 8});
    extract_cp2dt_hip(1, 0, d3, d3, e2, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, false, nullptr, 256, true, false, 0);
    ftk::lattice d4({2, 2, 2, 0}, {4, 4, 4, 1}), e3({0, 0, 0}, {8, 8, 8});
    extract_cp3dt_hip(1, 0, d4, d4, e3, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 256, true, true);
  }
  return 0;
}
