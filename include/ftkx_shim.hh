// C++ shim with the signatures of the reference's accelerator entry points, forwarding to the C ABI of include/ftkx.h:
//   extract_cp2dt_cuda / extract_cp2dt_sycl   include/ftk/filters/critical_point_tracker_2d_regular.hh:33-63
//   extract_cp3dt_cuda                        include/ftk/filters/critical_point_tracker_3d_regular.hh:42-56
// Header-only and templated on the lattice and record types, so that it needs no FTK header itself: FTK instantiates it with
// ftk::lattice (include/ftk/mesh/lattice.hh:16-69: nd(), start(i), size(i)) and ftk::feature_point_lite_t
// (include/ftk/features/feature_point_lite.hh:8-15); this repo's tests also instantiate it with ftkx::lattice / ftkx_cp_t.
// The argument list is the reference's, followed by what the CPU path knows and the CUDA path ignores (scaling factor,
// Jacobian symmetry, type filter); see INTEGRATION.md for the call-site patch.  Errors throw (the reference calls fatal()).
#ifndef FTKX_SHIM_HH
#define FTKX_SHIM_HH

#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "ftkx.h"

namespace ftkx {

namespace detail {
template <class Lattice>
inline void unpack(const Lattice &l, long long *st, long long *sz, size_t n)
{
  for (size_t i = 0; i < n; i ++) { st[i] = i < l.nd() ? (long long)l.start(i) : 0; sz[i] = i < l.nd() ? (long long)l.size(i) : 1; }
}
template <class PointLite>
inline std::vector<PointLite> take(int rc, ftkx_cp_t *recs, size_t n)
{
  static_assert(sizeof(PointLite) == sizeof(ftkx_cp_t), "record layouts must match (72 bytes, feature_point_lite.hh:8-15)");
  if (rc != FTKX_OK) {
    char msg[512] = {0};
    ftkx_last_error(nullptr, msg, sizeof msg);
    ftkx_free(recs);
    throw std::runtime_error(std::string("ftkx: ") + msg);
  }
  std::vector<PointLite> out(n);
  if (n) std::memcpy(static_cast<void *>(out.data()), recs, n * sizeof(ftkx_cp_t));
  ftkx_free(recs);
  return out;
}
}  // namespace detail

// 2D+t.  domain = {x0, y0, 0} x {nx, ny, INT_MAX} (vertex validity box), core = {x0, y0, t} x {nx, ny, 1} (corners to enumerate),
// ext = {0, 0} x {DW, DH} (array extents), exactly as the call sites build them (2d:333-347) -- but with the CPU path's
// domain sizes, not size - 1.  Record tags are work indices inside `core`, as the caller expects.
template <class PointLite, class Lattice>
std::vector<PointLite> extract_cp2dt_hip(
    int scope, int current_timestep, const Lattice &domain, const Lattice &core, const Lattice &ext,
    const double *Vc, const double *Vn, const double *Jc, const double *Jn, const double *Sc, const double *Sn,
    bool use_explicit_coords, const double *coords,
    unsigned long long vector_field_scaling_factor, bool jacobian_symmetric = false,
    bool use_type_filter = false, unsigned int type_filter = 0, int device = 0)
{
  long long dst[3], dsz[3], cst[3], csz[3], est[2], esz[2];
  detail::unpack(domain, dst, dsz, 3); detail::unpack(core, cst, csz, 3); detail::unpack(ext, est, esz, 2);
  ftkx_options o;
  ftkx_default_options(&o);
  o.jacobian_symmetric = jacobian_symmetric; o.use_type_filter = use_type_filter; o.type_filter = type_filter;
  o.tag_mode = FTKX_TAG_WORK_INDEX;                  // what from_work_index() at 2d:390, 419 expects
  ftkx_cp_t *recs = nullptr;
  size_t n = 0;
  const int rc = ftkx_extract_cp2dt(scope, current_timestep, dst, dsz, cst, csz, est, esz, Vc, Vn, Jc, Jn, Sc, Sn,
                                    use_explicit_coords ? 1 : 0, coords, vector_field_scaling_factor, &o, device, &recs, &n);
  return detail::take<PointLite>(rc, recs, n);
}

// 3D+t.  domain4 / core4 are 4-dimensional, ext3 3-dimensional (3d:205-247).
template <class PointLite, class Lattice>
std::vector<PointLite> extract_cp3dt_hip(
    int scope, int current_timestep, const Lattice &domain4, const Lattice &core4, const Lattice &ext3,
    const double *Vc, const double *Vl, const double *Jc, const double *Jl, const double *Sc, const double *Sl,
    unsigned long long vector_field_scaling_factor, bool jacobian_symmetric = false, bool robust = true, int device = 0)
{
  long long dst[4], dsz[4], cst[4], csz[4], est[3], esz[3];
  detail::unpack(domain4, dst, dsz, 4); detail::unpack(core4, cst, csz, 4); detail::unpack(ext3, est, esz, 3);
  ftkx_options o;
  ftkx_default_options(&o);
  o.jacobian_symmetric = jacobian_symmetric; o.robust = robust;
  o.tag_mode = FTKX_TAG_WORK_INDEX;
  ftkx_cp_t *recs = nullptr;
  size_t n = 0;
  const int rc = ftkx_extract_cp3dt(scope, current_timestep, dst, dsz, cst, csz, est, esz, Vc, Vl, Jc, Jl, Sc, Sl,
                                    vector_field_scaling_factor, &o, device, &recs, &n);
  return detail::take<PointLite>(rc, recs, n);
}

}  // namespace ftkx
#endif
