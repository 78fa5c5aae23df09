// C++ host side above the C ABI: a tracker with the reference's operator interface for the sweep path.
//
// Mirrors, method for method, the part of
//   ftk::critical_point_tracker_regular / critical_point_tracker_{2d,3d}_regular
//   (include/ftk/filters/critical_point_tracker.hh:27-186, regular_tracker.hh:19-98,
//    critical_point_tracker_2d_regular.hh:107-141, critical_point_tracker_3d_regular.hh:60-92)
// that drives the simplex sweep: set_domain / set_array_domain / set_*_field_source / set_jacobian_symmetric / initialize /
// push_{scalar,vector}_field_snapshot / advance_timestep / update_timestep / get_critical_points.
// Same names, same argument meaning, same call order as python/pyftk.cpp:93-142 and filters/json_interface.hh:606-725 use.
// The sweep itself runs in the HIP kernels behind include/ftkx.h; there is no CPU implementation behind this class.
// finalize() runs pass 2 (trace_critical_points_offline + cc2curves) on the host through ftkx_trace_curves; post_process() is
// json_interface::post_process with its default options (ftkx_post_process_curves); the write_/read_ members are the
// reference's record-stream formats (ftkx_write_critical_points & co.).
#ifndef FTKX_TRACKER_HH
#define FTKX_TRACKER_HH

#include <cstddef>
#include <limits>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "ftkx.h"
#include "ftkx_slab.h"

namespace ftkx {

// spatial lattice: starts + sizes, like ftk::lattice (include/ftk/mesh/lattice.hh:16-69)
struct lattice {
  std::vector<long long> starts, sizes;
  lattice() {}
  lattice(const std::vector<long long> &st, const std::vector<long long> &sz) : starts(st), sizes(sz) {}
  size_t nd() const { return sizes.size(); }
  long long start(size_t i) const { return starts[i]; }
  long long size(size_t i) const { return sizes[i]; }
  long long upper_bound(size_t i) const { return starts[i] + sizes[i] - 1; }
};

enum { SOURCE_NONE = 0, SOURCE_GIVEN = 1, SOURCE_DERIVED = 2 };   // include/ftk/filters/critical_point_tracker.hh:19-24

// == ftk::feature_point_t restricted to what the sweep fills (include/ftk/features/feature_point.hh:128-139)
struct feature_point_t {
  double x[3] = {0, 0, 0};
  double t = 0;
  int timestep = 0;
  double scalar[3] = {0, 0, 0};
  unsigned int type = 0;
  bool ordinal = false;
  unsigned long long tag = 0;
};

// Order of the reference's std::map<element_t, feature_point_t>: simplicial_regular_mesh_element::operator<
// (mesh/simplicial_regular_mesh.hh:327-337) = corner as a vector (x first, ..., time last), then the type -- expressed on
// element tags (tag = linear corner index * ntypes + type, simplicial_regular_mesh.hh:496-502).  Tags that wrapped in
// FTKX_TAG_REFERENCE mode on very large meshes still get a strict weak order, but not the reference's.
struct element_order {
  long long n[3] = {1, 1, 1};   // spatial sizes of the mesh (= domain sizes)
  int nd = 2;
  bool operator()(unsigned long long a, unsigned long long b) const
  {
    if (a == b) return false;
    const unsigned long long ntypes = nd == 2 ? 12 : 60;
    unsigned long long ia = a / ntypes, ib = b / ntypes;
    for (int d = 0; d < nd; d ++) {
      const unsigned long long m = (unsigned long long)(n[d] > 0 ? n[d] : 1);
      const unsigned long long ca = ia % m, cb = ib % m;
      if (ca != cb) return ca < cb;
      ia /= m; ib /= m;
    }
    if (ia != ib) return ia < ib;                  // time
    return a % ntypes < b % ntypes;
  }
  // the same order as one comparison of integers: (spatial index with x most significant, time * ntypes + type)
  std::pair<unsigned long long, unsigned long long> key(unsigned long long tag) const
  {
    const unsigned long long ntypes = nd == 2 ? 12 : 60;
    unsigned long long i = tag / ntypes, spatial = 0;
    for (int d = 0; d < nd; d ++) {
      const unsigned long long m = (unsigned long long)(n[d] > 0 ? n[d] : 1);
      spatial = spatial * m + i % m;
      i /= m;
    }
    return {spatial, i * ntypes + tag % ntypes};
  }
};
typedef std::map<unsigned long long, feature_point_t, element_order> discrete_map_t;

struct ftkx_error : public std::runtime_error {
  int code;
  ftkx_error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

class critical_point_tracker_regular {
public:
  critical_point_tracker_regular(int nd /*2|3*/, int device_id = 0);
  // Several GPUs behind ONE tracker (the reference keeps a device list on the filter, filter.hh:47-61, and partitions inside the
  // tracker, regular_tracker.hh:127-149): timesteps are dealt to the devices in blocks of `block` consecutive steps, a snapshot
  // goes to the device(s) whose steps read it (the first slice of a block also to the previous block's device: the t-slab halo),
  // every device has its own context and host thread, update_timestep() only queues the step -- the devices sweep different
  // timesteps at the same time -- and the getters / finalize() wait for the queues.  The quantisation factor stays the
  // reference's sequential sticky minimum: each step waits for the reductions of all earlier slices (computed on whichever
  // device holds them), never for their sweeps.  A device may be listed more than once (tests on a one-GPU box).
  critical_point_tracker_regular(int nd, const std::vector<int> &device_ids, int block = 2);
  virtual ~critical_point_tracker_regular();
  critical_point_tracker_regular(const critical_point_tracker_regular &) = delete;
  critical_point_tracker_regular &operator=(const critical_point_tracker_regular &) = delete;

  // regular_tracker.hh:24-25
  void set_domain(const lattice &l) { domain = l; }
  void set_array_domain(const lattice &l) { array_domain = l; }
  // critical_point_tracker.hh:36-48
  void set_scalar_field_source(int s) { scalar_field_source = s; }
  void set_vector_field_source(int s) { vector_field_source = s; }
  void set_jacobian_field_source(int s) { jacobian_field_source = s; }
  void set_jacobian_symmetric(bool s) { is_jacobian_field_symmetric = s; }
  void set_enable_robust_detection(bool b) { enable_robust_detection = b; }
  void set_enable_computing_degrees(bool b) { enable_computing_degrees = b; }
  // critical_point_tracker.hh:38: trajectories grow after every interval sweep (trace_critical_points_online), the discrete
  // points are consumed, finalize() only hands the trajectories over.  Single-device trackers.
  void set_enable_streaming_trajectories(bool b) { enable_streaming_trajectories = b; }
  void set_type_filter(unsigned int f) { use_type_filter = true; type_filter = f; }
  // regular_tracker.hh:38 -- REGULAR_COORDS_BOUNDS: x0,x1,y0,y1[,z0,z1]
  void set_coords_bounds(const std::vector<double> &b) { bounds_coords = b; mode_phys_coords = 1; }
  // regular_tracker.hh:39-40 -- REGULAR_COORDS_RECTILINEAR: one coordinate array per axis, indexed by the vertex coordinate;
  // REGULAR_COORDS_EXPLICIT: ndarray (ncomp, n0, n1), first index fastest
  void set_coords_rectilinear(const std::vector<std::vector<double>> &c) { rectilinear_coords = c; mode_phys_coords = 2; }
  void set_coords_explicit(const double *c, int ncomp, size_t n0, size_t n1)
  { explicit_coords.assign(c, c + (size_t)ncomp * n0 * n1); explicit_ncomp = ncomp; explicit_n0 = n0; explicit_n1 = n1; mode_phys_coords = 3; }
  // tracker.hh:40-41
  void set_current_timestep(int t) { current_timestep = t; if (field_data_snapshots.empty()) next_push_timestep = t; }   // the next snapshot pushed is timestep t
  int get_current_timestep() const { return current_timestep; }
  // not in the reference: knobs of this implementation
  void set_exact_only(bool b) { exact_only = b; }          // never cull (every simplex takes the integer test)
  void set_tag_mode(int m) { tag_mode = m; }               // FTKX_TAG_REFERENCE (default) | FTKX_TAG_EXACT64
  // Deferred collection (single-device trackers without streaming trajectories): update_timestep() QUEUES the step's sweep and
  // collects the step before it -- the device works on step t + 1 (continuing on the device from step t's running minimum) while the
  // host takes step t's records.  Same records, factors and statistics; they become visible one step later, and every accessor,
  // sync() and finalize() collect what is still out first.  A snapshot pushed as a DEVICE pointer is borrowed, not copied.  Up to THREE
  // steps are in flight (a step is collected when the second step after it is advanced): with this on a borrowed pointer must stay valid
  // until TWO further steps after the one that pops it have been advanced -- snapshot t is popped by advance_timestep() number t and is
  // free when advance_timestep() number t + 2 has returned -- or until sync() has been called.
  // depth > 1: the sweeps of `depth` consecutive steps are queued as ONE device-driven pass (ftkx_sweep_series_submit with n = depth: one
  // mask launch over the batch's new snapshots, one tail) when the last of them is advanced -- the latency chain behind the mask kernel
  // is paid once per batch; a popped snapshot stays resident (and a borrowed device pointer must stay valid) until its batch has been
  // collected, which is when the second batch after it is submitted: 3 * depth + 2 steps after its push at most (ftk_amd.Tracker keeps
  // borrowed tensors alive for exactly that long).  Same records, factors and statistics; sync(), finalize() and every accessor submit a
  // partial batch and collect everything, after which nothing is borrowed any more.
  void set_deferred_collection(bool b, int depth = 1) { sync(); deferred_collection = b; deferred_depth = depth > 1 ? depth : 1; }
  // Several RANKS behind the tracker -- one process per GPU, or one tracker per device and thread in one process.  The reference keeps an
  // MPI communicator on the filter and distributes inside the tracker (regular_tracker.hh:127-149), gathering the discrete points on the
  // root in front of pass 2 (critical_point_tracker.hh:689).  Here the series of `nt` timesteps is cut in TIME (include/ftkx_slab.h): this
  // rank pushes the snapshots of ITS slab -- ftkx_slab_range(nt, nranks, rank), the first push is timestep t0 -- and calls
  // advance_timestep() / update_timestep() as usual, which only RECORD the steps; the slab is swept as one device-driven pass (masks,
  // factors across slabs, the compact halo and the records queued at once: ftkx_slab_submit / _complete) when results are first asked for
  // (sync(), a getter, finalize()), and finalize() gathers every rank's points on rank 0, which traces them (the other ranks end without
  // trajectories, like the reference's non-root ranks).  sync() / finalize() are collective.  Snapshots stay resident until then: device
  // pointers are borrowed for that long.  Call after the constructor, before the first push; one of:
  void set_communicator(void *nccl_comm, int rank, int nranks, int nt);                  // RCCL: ncclComm_t over the ranks
  void set_slab_transport(const ftkx_slab_transport &tr, int rank, int nranks, int nt);  // the caller's own transport
  void set_slab_hub(ftkx_slab_hub *hub, int rank, int nt);                               // ranks of one process (ftkx_slab_hub_create)
  bool slab_mode() const { return slab != nullptr; }
  const ftkx_slab *slab_host() const { return slab; }
  void set_stream(void *hip_stream);

  void initialize();                                        // regular_tracker.hh:105-149 (single rank: local == global)
  void reset();                                             // critical_point_tracker_2d_regular.hh:227-236

  // host pointers, reference ndarray layout (first index fastest).  device = true: device pointers, adopted without a copy
  // and owned by the caller until the snapshot is popped.
  void push_scalar_field_snapshot(const double *scalar, bool device = false);               // 2d:238-250, 3d:125-137
  void push_vector_field_snapshot(const double *vector, bool device = false);               // 2d:252-261, 3d:139-148
  void push_field_data_snapshot(const double *scalar, const double *vector, const double *jacobian, bool device = false);  // critical_point_tracker.hh:137-140
  bool pop_field_data_snapshot();

  bool advance_timestep();                                  // critical_point_tracker.hh:841-848
  void update_timestep();                                   // 2d:263-433, 3d:150-308 -- THE SWEEP

  void finalize();                                          // 2d:143-225, 3d:86-117 -> trace_critical_points_offline
  // traced curves after finalize(): each an ordered list of points (feature_curve_t), loop flag alongside
  const std::vector<std::vector<feature_point_t>> &get_traced_critical_points() const;      // (one vector per curve: built from the flat form when asked for)
  // the traced curves as finalize() / post_process() keep them: the points of all curves one after the other, curve c = [offsets[c], offsets[c + 1])
  const std::vector<feature_point_t> &get_traced_points() const { return traced_points; }
  const std::vector<long long> &get_traced_offsets() const { return traced_offsets; }
  size_t num_traced_curves() const { return traced_offsets.size() - 1; }
  const std::vector<int> &get_traced_loop_flags() const { return traced_loop; }
  const std::vector<int> &get_traced_ids() const { return traced_id; }   // label of each curve in the reference's multimap
  // json_interface::post_process, default options (filters/json_interface.hh:758-800): smooth types, rotate, split_all,
  // reorder, adjust_time.  Rewrites the traced curves in place.
  void post_process();

  // i/o of discrete points (critical_point_tracker.hh:104-116) and of traced curves (:118-131), the reference's formats
  void write_critical_points_json(const std::string &filename) const;
  void write_critical_points_binary(const std::string &filename) const;
  void write_critical_points_text(const std::string &filename) const;
  void read_critical_points_json(const std::string &filename);       // -> put_critical_points
  void read_critical_points_binary(const std::string &filename);
  void put_critical_points(const std::vector<feature_point_t> &);     // critical_point_tracker_regular.hh:40-46
  void write_traced_critical_points_json(const std::string &filename) const;
  void write_traced_critical_points_binary(const std::string &filename) const;
  void write_traced_critical_points_text(const std::string &filename) const;

  std::vector<feature_point_t> get_critical_points() const; // critical_point_tracker_regular.hh:32-38 (sorted by element)
  const discrete_map_t &get_discrete_critical_points() const;   // (the std::map view: built from the flat store when asked for)
  size_t num_discrete_critical_points() const { sync(); return points.size(); }

  void sync() const;                                        // multi-device: wait for every queued step (no-op with one device)
  int num_devices() const;
  unsigned long long get_vector_field_scaling_factor() const { sync(); return vector_field_scaling_factor; }
  double get_vector_field_resolution() const { sync(); return vector_field_resolution; }
  ftkx_stats get_last_stats() const { sync(); return last_stats; }
  ftkx_ctx *context() { return ctx; }

protected:
  void update_vector_field_scaling_factor(int minbits = 8, int maxbits = 21);   // critical_point_tracker.hh:850-864
  void check(int rc) const;

  int nd;
  ftkx_ctx *ctx = nullptr;
  lattice domain, array_domain, local_domain, local_array_domain;
  int scalar_field_source = SOURCE_NONE, vector_field_source = SOURCE_NONE, jacobian_field_source = SOURCE_NONE;
  bool is_jacobian_field_symmetric = false;
  bool enable_robust_detection = true, enable_computing_degrees = false, enable_streaming_trajectories = false;
  bool deferred_collection = false;
  mutable std::vector<int> open_steps;                      // deferred collection: the timesteps of the sweeps that are queued and not yet collected (at most three)
  void collect_open_step() const;
  int deferred_depth = 1;
  mutable std::vector<int> batch_ts, batch_scopes, batch_drops;    // deferred collection in batches: the steps recorded and not yet queued; the snapshots popped meanwhile
  void submit_batch() const;
  ftkx_online_tracer *online = nullptr;
  void grow();                                                 // 2d:288-322: trace_critical_points_online on the points found since the last call
  bool use_type_filter = false;
  unsigned int type_filter = 0;
  bool exact_only = false;
  int tag_mode = FTKX_TAG_REFERENCE;
  int mode_phys_coords = 0;
  std::vector<double> bounds_coords;
  std::vector<std::vector<double>> rectilinear_coords;
  std::vector<double> explicit_coords;
  int explicit_ncomp = 0;
  size_t explicit_n0 = 0, explicit_n1 = 0;
  bool initialized = false;

  int current_timestep = 0;
  int result_timestep = -1;                                 // multi-device: the timestep last_stats / the scaling factor belong to
  std::vector<int> field_data_snapshots;                    // timesteps resident on the device (<= 2, a deque in the reference)
  int next_push_timestep = 0;
  double vector_field_resolution = std::numeric_limits<double>::max();   // sticky running minimum (never reset)
  unsigned long long vector_field_scaling_factor = 1;
  // The discrete points (the reference keeps a std::map<element_t, feature_point_t>): a flat array in the reference's element order with
  // unique tags (`points`, its order keys beside it), what the sweeps have delivered since it was last brought up to date
  // (`pending_points`: a streaming run delivers ~10^3 points per step), and -- only when somebody asks for it -- the std::map view.
  // Sweeps in time order with 64-bit tags deliver strictly ascending tags (`pending_ascending`): finalize() then traces the pending
  // points as they are, without ordering anything.
  mutable std::vector<feature_point_t> points;
  mutable std::vector<std::pair<unsigned long long, unsigned long long>> point_keys;
  mutable std::vector<feature_point_t> pending_points;
  mutable bool pending_ascending = true;
  mutable discrete_map_t discrete_critical_points;
  mutable bool map_valid = true;
  element_order order_;
  void wait_devices() const;
  void flush_points() const;
  std::vector<feature_point_t> traced_points;
  std::vector<long long> traced_offsets = std::vector<long long>(1, 0);
  mutable std::vector<std::vector<feature_point_t>> traced_critical_points;
  mutable bool traced_nested_valid = true;
  std::vector<int> traced_loop, traced_id;
  ftkx_stats last_stats;

  // slab mode (set_communicator / set_slab_transport / set_slab_hub): the C++ host of this rank's slab, the steps recorded so far
  ftkx_slab *slab = nullptr;
  int slab_nt = 0, slab_rank = 0, slab_nranks = 1, slab_t0 = 0, slab_t1 = 0;
  mutable std::vector<int> slab_steps;
  mutable bool slab_swept = false;
  void enter_slab_mode(int rank, int nranks, int nt);
  void run_slab() const;
  struct multi_engine;                                        // per-device contexts + worker threads (tracker.cpp); null with one device
  std::unique_ptr<multi_engine> multi;
  void apply_configuration(ftkx_ctx *c);                      // mesh, options, coordinates of initialize() on one context
  void take_records(const ftkx_cp_t *recs, size_t n, int timestep);
  void push_everywhere(int kind, int t, const double *s, const double *v, const double *j, bool device);   // one snapshot -> the context(s) that read it
  void write_discrete(const std::string &filename, int format) const;
  void read_discrete(const std::string &filename, int format);
  void write_traced(const std::string &filename, int format) const;
};

struct critical_point_tracker_2d_regular : public critical_point_tracker_regular {
  explicit critical_point_tracker_2d_regular(int device_id = 0) : critical_point_tracker_regular(2, device_id) {}
};
struct critical_point_tracker_3d_regular : public critical_point_tracker_regular {
  explicit critical_point_tracker_3d_regular(int device_id = 0) : critical_point_tracker_regular(3, device_id) {}
};

}  // namespace ftkx

// C handles of the same class for non-C++ callers (ctypes, cgo, JNI ...).  Errors come back as FTKX_E_* codes.
extern "C" {
typedef struct ftkx_tracker ftkx_tracker;
int  ftkx_tracker_create(ftkx_tracker **out, int nd, int device_id);
/* one tracker over several devices (see the C++ constructor above); block = consecutive timesteps per device (>= 1) */
int  ftkx_tracker_create_multi(ftkx_tracker **out, int nd, const int *device_ids, int ndev, int block);
int  ftkx_tracker_sync(ftkx_tracker *);
void ftkx_tracker_destroy(ftkx_tracker *);
int  ftkx_tracker_last_error(const ftkx_tracker *, char *buf, size_t n);
int  ftkx_tracker_set_domain(ftkx_tracker *, const long long *starts, const long long *sizes);
int  ftkx_tracker_set_array_domain(ftkx_tracker *, const long long *starts, const long long *sizes);
int  ftkx_tracker_set_sources(ftkx_tracker *, int scalar, int vector, int jacobian, int jacobian_symmetric);
int  ftkx_tracker_set_flags(ftkx_tracker *, int robust, int use_type_filter, unsigned type_filter, int compute_degrees, int exact_only, int tag_mode);
int  ftkx_tracker_set_stream(ftkx_tracker *, void *hip_stream);
int  ftkx_tracker_set_current_timestep(ftkx_tracker *, int t);
int  ftkx_tracker_set_deferred_collection(ftkx_tracker *, int on);   /* not in the reference: see critical_point_tracker_regular::set_deferred_collection; on > 1: batches of `on` steps */
/* several ranks behind the tracker (critical_point_tracker_regular::set_communicator / set_slab_transport / set_slab_hub): this rank's
 * tracker takes the snapshots of its timestep slab, sweeps it as one device-driven pass, and ftkx_tracker_finalize gathers the points on rank 0 */
int  ftkx_tracker_set_communicator(ftkx_tracker *, void *nccl_comm, int rank, int nranks, int nt);
int  ftkx_tracker_set_slab_transport(ftkx_tracker *, const ftkx_slab_transport *tr, int rank, int nranks, int nt);
int  ftkx_tracker_set_slab_hub(ftkx_tracker *, ftkx_slab_hub *hub, int rank, int nt);
int  ftkx_tracker_set_enable_streaming_trajectories(ftkx_tracker *, int on);   /* critical_point_tracker.hh:38; before the first step */   /* tracker::set_current_timestep (filters/tracker.hh:40), before the first push */
int  ftkx_tracker_set_coords_bounds(ftkx_tracker *, const double *bounds /* 2*nd values */);
int  ftkx_tracker_set_coords_rectilinear(ftkx_tracker *, const double *x, size_t nx, const double *y, size_t ny, const double *z, size_t nz);
int  ftkx_tracker_set_coords_explicit(ftkx_tracker *, const double *coords, int ncomp, size_t n0, size_t n1);
int  ftkx_tracker_initialize(ftkx_tracker *);
int  ftkx_tracker_push_scalar_field_snapshot(ftkx_tracker *, const double *s, int on_device);
int  ftkx_tracker_push_vector_field_snapshot(ftkx_tracker *, const double *v, int on_device);
int  ftkx_tracker_push_field_data_snapshot(ftkx_tracker *, const double *s, const double *v, const double *j, int on_device);
int  ftkx_tracker_advance_timestep(ftkx_tracker *);
int  ftkx_tracker_update_timestep(ftkx_tracker *);
int  ftkx_tracker_num_critical_points(const ftkx_tracker *, size_t *n);
/* copies up to cap records; ordinal[i], timestep[i] nullable */
int  ftkx_tracker_get_critical_points(const ftkx_tracker *, ftkx_cp_t *out, int *ordinal, int *timestep, size_t cap);
int  ftkx_tracker_get_scaling(const ftkx_tracker *, unsigned long long *factor, double *resolution);
int  ftkx_tracker_get_stats(const ftkx_tracker *, ftkx_stats *st);
int  ftkx_tracker_finalize(ftkx_tracker *);
/* after finalize: number of curves / total points; then offsets[n_curves+1], tags[n_points] (element tags in curve order), loop[n_curves] */
int  ftkx_tracker_num_curves(const ftkx_tracker *, size_t *n_curves, size_t *n_points);
int  ftkx_tracker_get_curves(const ftkx_tracker *, long long *offsets, unsigned long long *tags, int *loop);
/* json_interface::post_process defaults on the traced curves; afterwards get_curves returns the trajectories, and
 * ftkx_tracker_get_curve_points the per-point (smoothed) type and (adjusted) time, ids[n_curves] the multimap labels */
int  ftkx_tracker_post_process(ftkx_tracker *);
int  ftkx_tracker_get_curve_points(const ftkx_tracker *, unsigned int *type, double *t, int *ids);
/* format = FTKX_FORMAT_*; traced = 0 discrete points (write_critical_points_*), 1 traced curves (write_traced_critical_points_*) */
int  ftkx_tracker_write(const ftkx_tracker *, const char *path, int format, int traced);
int  ftkx_tracker_read_critical_points(ftkx_tracker *, const char *path, int format);
}
#endif
