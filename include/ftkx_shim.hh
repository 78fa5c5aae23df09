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


// ---- the RESIDENT form: what the patched tracker uses by default (patches/ftk-xl-hip.patch) -------------------
// The one-shot calls above cross PCIe with V, J and S of both snapshots on every call (104 bytes per vertex and slice at 3D), because that
// is what the reference boundary hands over (2d:369-384).  A tracker that owns a resident_sweep hands each snapshot over ONCE, as the
// caller gave it -- 8 bytes per vertex for scalar input, V = gradient and J derived on the device bit for bit like ndarray/grad.hh --
// keeps it in HBM for the two steps that read it, and gets the sticky scaling factor (critical_point_tracker.hh:850-864) formed on the
// device between the mask kernel and the exact test.  The class mirrors std::deque<field_data_snapshot_t> (critical_point_tracker.hh:155-159):
// push_* = emplace_back, pop_front = pop_front, sweep = the two extract_* calls of one update_timestep() (2d:263-433, 3d:150-308).
// Header-only over the C ABI; errors throw like the calls above.
class resident_sweep {
 public:
  resident_sweep() = default;
  resident_sweep(const resident_sweep &) = delete;
  resident_sweep &operator=(const resident_sweep &) = delete;
  ~resident_sweep() { close(); }

  bool active() const { return ctx_ != nullptr; }
  size_t size() const { return n_; }                 // snapshots resident
  int front_timestep() const { return t0_; }         // timestep of the oldest one (valid while size() > 0)
  ftkx_ctx *context() const { return ctx_; }

  // spatial lattices as the tracker holds them (regular_tracker.hh:61-62): domain (vertex validity box, the CPU path's: size, not size - 1),
  // core = local_domain (corners to enumerate), array extents = the dims of the pushed ndarray (2d:364-366)
  template <class Lattice>
  void open(int nd, int device, const Lattice &domain, const Lattice &core, const size_t *array_dims)
  {
    close();
    check(ftkx_create(&ctx_, nd, device));
    nd_ = nd;
    long long dst[3] = {0, 0, 0}, dsz[3] = {1, 1, 1}, cst[3] = {0, 0, 0}, csz[3] = {1, 1, 1}, est[3] = {0, 0, 0}, esz[3] = {1, 1, 1};
    for (int d = 0; d < nd; d ++) {
      dst[d] = (long long)domain.start(d); dsz[d] = (long long)domain.size(d);
      cst[d] = (long long)core.start(d); csz[d] = (long long)core.size(d);
      esz[d] = (long long)array_dims[d];
    }
    check(ftkx_set_mesh(ctx_, dst, dsz, cst, csz, est, esz));
  }
  void close() { if (ctx_) ftkx_destroy(ctx_); ctx_ = nullptr; n_ = 0; }

  // tag_mode is forced to FTKX_TAG_WORK_INDEX: what from_work_index() at the call sites expects (2d:387-395)
  void set_options(ftkx_options o)
  {
    o.tag_mode = FTKX_TAG_WORK_INDEX;
    if (have_opt_ && std::memcmp(&o, &opt_, sizeof o) == 0) return;
    check(ftkx_set_options(ctx_, &o));
    opt_ = o; have_opt_ = true;
  }
  void set_coords_rectilinear(const double *x, size_t nx, const double *y, size_t ny, const double *z, size_t nz) { check(ftkx_set_coords_rectilinear(ctx_, x, nx, y, ny, z, nz)); }
  void set_coords_explicit(const double *coords, int ncomp, size_t n0, size_t n1) { check(ftkx_set_coords_explicit(ctx_, coords, ncomp, n0, n1)); }

  // host arrays of the caller (ndarray<double>::data()), copied to the device before the call returns
  void push_scalar(int t, const double *S) { note(t); check(ftkx_push_scalar_slice(ctx_, t, S, 0)); n_ ++; }
  void push(int t, const double *V, const double *J, const double *S) { note(t); check(ftkx_push_slice(ctx_, t, V, J, S, 0)); n_ ++; }
  bool pop_front()
  {
    if (!ctx_ || n_ == 0) return false;
    check(ftkx_drop_slice(ctx_, t0_));
    t0_ ++; n_ --;
    return true;
  }
  void clear() { while (pop_front()) {} }

  // One update_timestep(): the ordinal sweep of slice t and, if `interval`, the sweep of [t, t + 1], as ONE pass under
  //   factor = scaling_factor(min(running_resolution, resolution(V_t), resolution(V_t+1)))
  // -- update_vector_field_scaling_factor runs over every snapshot in the deque before either sweep (2d:267-269, 3d:156-158).
  // running_resolution: the tracker's sticky vector_field_resolution, in and out.  Tags are work indices inside the step's core, each
  // inside its own scope (from_work_index, simplicial_regular_mesh.hh:480-493); the records' aux word says which scope.
  template <class PointLite>
  void sweep(int t, bool interval, double &running_resolution, unsigned long long &factor, std::vector<PointLite> &ordinal_out, std::vector<PointLite> &interval_out)
  {
    static_assert(sizeof(PointLite) == sizeof(ftkx_cp_t), "record layouts must match (72 bytes, feature_point_lite.hh:8-15)");
    if (!ctx_ || n_ == 0 || t != t0_ || (interval && n_ < 2))
      throw std::runtime_error("ftkx: resident sweep of timestep " + std::to_string(t) + ": the resident snapshots start at " + std::to_string(t0_) + " (" + std::to_string(n_) + " of them)");
    ordinal_out.clear(); interval_out.clear();
    const ftkx_cp_t *recs = nullptr;
    size_t n = 0;
    unsigned long long f = 0;
    const int scope = interval ? FTKX_SCOPE_BOTH : FTKX_SCOPE_ORDINAL;
    check(ftkx_sweep_series(ctx_, &t, &scope, 1, &running_resolution, &f, &recs, &n));
    size_t n_ordinal = 0;
    for (size_t i = 0; i < n; i ++) n_ordinal += (size_t)ftkx_cp_ordinal(recs + i);
    ordinal_out.reserve(n_ordinal); interval_out.reserve(n - n_ordinal);
    for (size_t i = 0; i < n; i ++) {
      PointLite p;
      std::memcpy(static_cast<void *>(&p), recs + i, sizeof(ftkx_cp_t));
      (ftkx_cp_ordinal(recs + i) ? ordinal_out : interval_out).push_back(p);
    }
    factor = f;
  }

 private:
  void check(int rc) const
  {
    if (rc == FTKX_OK) return;
    char msg[512] = {0};
    ftkx_last_error(ctx_, msg, sizeof msg);
    throw std::runtime_error(std::string("ftkx: ") + msg);
  }
  void note(int t)
  {
    if (!ctx_) throw std::runtime_error("ftkx: resident_sweep::push before open()");
    if (n_ == 0) t0_ = t;
    else if (t != t0_ + (int)n_) throw std::runtime_error("ftkx: resident snapshots must be pushed in timestep order");
  }
  ftkx_ctx *ctx_ = nullptr;
  int nd_ = 0, t0_ = 0;
  size_t n_ = 0;
  ftkx_options opt_;
  bool have_opt_ = false;
};

}  // namespace ftkx
#endif
