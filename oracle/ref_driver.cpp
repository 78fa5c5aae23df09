// TEST INFRASTRUCTURE -- not part of the product.
//
// Driver that runs the *real* reference CPU sweep (hguo/ftk headers, compiled where they
// lie under /root/reference) and dumps what the parity suite needs:
//   - the per-timestep input field exactly as the tracker API received it,
//   - the quantisation factor in force at every sweep,
//   - the discrete critical-point record stream (std::map order = sorted by element).
//
// It drives the tracker the way the reference's own callers do:
//   python/pyftk.cpp:93-142 and include/ftk/filters/json_interface.hh:606-725
//   (set_*_field_source / set_domain / initialize / push / advance_timestep / update_timestep).
// It contains no restatement of the algorithm: everything numerical comes from
// /root/reference/include.  Built only by oracle/Makefile into oracle/_ref/ (git-ignored).
//
// usage:
//   ftk_ref_driver synthetic <name> <DW> <DH> <DD> <DT> <out.bin> [x0 x0 x0 dir dir dir] [nthreads]
//   ftk_ref_driver file <in.bin> <out.bin> [nthreads]      (in.bin: see read_input())
//   ftk_ref_driver time <in.bin> [nthreads]                (prints sweep seconds, no dump)
//   ftk_ref_driver tables <out.txt>                         (dumps the unit-simplex tables)
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <string>
#include <vector>
#include <ftk/ndarray/synthetic.hh>
#include <ftk/filters/critical_point_tracker_2d_regular.hh>
#include <ftk/filters/critical_point_tracker_3d_regular.hh>

#ifdef FTKX_SHIM
// Built a second time as oracle/_ref/ftk_shim_driver, against the reference's headers WITH patches/ftk-xl-hip.patch applied (oracle/Makefile
// applies it to a scratch copy) and linked with the patch's new source file: the same REAL reference trackers, told use_accelerator("hip").
// Nothing is overridden here: update_timestep() is the reference's, its accelerator branch (critical_point_tracker_2d_regular.hh:332-429,
// ..._3d_regular.hh:203-304) calls libftkx.so through the patch's extract_cp{2,3}dt_hip, and the reference's own lattices, its own
// from_work_index / to_integer round trip (2d:387-395), its own std::map and its own finalize() consume the records.
#if !FTK_HAVE_HIP
#error "FTKX_SHIM: build against the patched reference headers (oracle/Makefile, target shim)"
#endif
#endif

struct input_t {
  int nd = 2, nv = 1;            // nv = 1: scalar field; nv = nd: vector field
  int D[3] = {1, 1, 1}, DT = 1;
  std::vector<std::vector<double>> steps;  // DT arrays of nv*D0*D1*D2 doubles (component fastest)
};

template <class Base>
struct probe : public Base {
  probe(diy::mpi::communicator comm) : Base(comm), ftk::tracker(comm) {}
  std::vector<uint64_t> factors;  // factor in force at each update_timestep()
  std::vector<int> factor_steps;
  double sweep_seconds = 0, push_seconds = 0;
  std::vector<double> push_ms, sweep_ms;   // per call
  // (push_*_field_snapshot: virtual in critical_point_tracker.hh:130-131.  Timed because that is where a snapshot costs the host: the
  // ndarray copies and gradient / jacobian of the CPU path, the upload of the patched FTK_XL_HIP path)
  void push_scalar_field_snapshot(const ftk::ndarray<double> &a) override {
    const auto t0 = std::chrono::high_resolution_clock::now();
    Base::push_scalar_field_snapshot(a);
    push_ms.push_back(1e3 * std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count());
    push_seconds += 1e-3 * push_ms.back();
  }
  void push_vector_field_snapshot(const ftk::ndarray<double> &a) override {
    const auto t0 = std::chrono::high_resolution_clock::now();
    Base::push_vector_field_snapshot(a);
    push_ms.push_back(1e3 * std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count());
    push_seconds += 1e-3 * push_ms.back();
  }
#ifdef FTKX_SHIM
  bool resident() const { return this->hip_is_resident(); }     // (the patch's: snapshots in HBM, critical_point_tracker_regular.hh)
#else
  bool resident() const { return false; }
#endif
  void update_timestep() override {
    const auto t0 = std::chrono::high_resolution_clock::now();
    Base::update_timestep();
    const auto t1 = std::chrono::high_resolution_clock::now();
    sweep_seconds += std::chrono::duration<double>(t1 - t0).count();
    sweep_ms.push_back(1e3 * std::chrono::duration<double>(t1 - t0).count());
    factors.push_back(this->vector_field_scaling_factor);
    factor_steps.push_back(this->current_timestep);
  }
};

static ftk::ndarray<double> make_array(const input_t &in, int k)
{
  ftk::ndarray<double> a;
  std::vector<size_t> shape;
  if (in.nv > 1) shape.push_back(in.nv);
  for (int i = 0; i < in.nd; i ++) shape.push_back(in.D[i]);
  a.reshape(shape);
  if (in.nv > 1) a.set_multicomponents();
  std::memcpy(a.data(), in.steps[k].data(), sizeof(double) * in.steps[k].size());
  return a;
}

static input_t synthesize(const std::string &name, int DW, int DH, int DD, int DT, const double *x0dir)
{
  input_t in;
  in.DT = DT;
  auto put = [&](const ftk::ndarray<double> &a) {
    in.steps.emplace_back(a.data(), a.data() + a.nelem());
  };
  if (name == "woven") {                       // stream.hh:1468-1477
    in.nd = 2; in.nv = 1; in.D[0] = DW; in.D[1] = DH;
    for (int k = 0; k < DT; k ++) {
      const double t = DT == 1 ? 0.0 : double(k) / (DT - 1);
      put(ftk::synthetic_woven_2D<double>(DW, DH, t));
    }
  } else if (name == "merger_2d") {            // stream.hh:1538-1541
    in.nd = 2; in.nv = 1; in.D[0] = DW; in.D[1] = DH;
    for (int k = 0; k < DT; k ++) put(ftk::synthetic_merger_2D<double>(DW, DH, double(k) * 0.1));
  } else if (name == "moving_extremum_2d") {   // stream.hh:1490-1497
    in.nd = 2; in.nv = 1; in.D[0] = DW; in.D[1] = DH;
    const double x0[2] = {x0dir[0], x0dir[1]}, dir[2] = {x0dir[3], x0dir[4]};
    for (int k = 0; k < DT; k ++)
      put(ftk::synthetic_moving_extremum<double, 2>({(size_t)DW, (size_t)DH}, x0, dir, double(k)));
  } else if (name == "moving_extremum_3d") {   // stream.hh:1510-1517
    in.nd = 3; in.nv = 1; in.D[0] = DW; in.D[1] = DH; in.D[2] = DD;
    const double x0[3] = {x0dir[0], x0dir[1], x0dir[2]}, dir[3] = {x0dir[3], x0dir[4], x0dir[5]};
    for (int k = 0; k < DT; k ++)
      put(ftk::synthetic_moving_extremum<double, 3>({(size_t)DW, (size_t)DH, (size_t)DD}, x0, dir, double(k)));
  } else if (name == "double_gyre") {          // stream.hh:1544-1556
    in.nd = 2; in.nv = 2; in.D[0] = DW; in.D[1] = DH;
    for (int k = 0; k < DT; k ++)
      put(ftk::synthetic_double_gyre<double>(DW, DH, k * 0.1, false, 0.1, M_PI * 2, 0.25));
  } else {
    fprintf(stderr, "unknown synthetic case %s\n", name.c_str());
    exit(2);
  }
  return in;
}

// in.bin: int32 nd, nv, D0, D1, D2, DT; then DT * nv*D0*D1*D2 doubles
static input_t read_input(const char *fn)
{
  input_t in;
  FILE *fp = fopen(fn, "rb");
  if (!fp) { perror(fn); exit(2); }
  int32_t h[6];
  if (fread(h, sizeof(int32_t), 6, fp) != 6) exit(2);
  in.nd = h[0]; in.nv = h[1]; in.D[0] = h[2]; in.D[1] = h[3]; in.D[2] = h[4]; in.DT = h[5];
  const size_t n = (size_t)in.nv * in.D[0] * in.D[1] * in.D[2];
  for (int k = 0; k < in.DT; k ++) {
    std::vector<double> a(n);
    if (fread(a.data(), sizeof(double), n, fp) != n) exit(2);
    in.steps.push_back(std::move(a));
  }
  fclose(fp);
  return in;
}

template <class Tracker>
static void run(const input_t &in, int nthreads, const char *out, bool robust, unsigned type_filter, bool degrees, const std::vector<double> &bounds)
{
  diy::mpi::communicator comm;
  probe<Tracker> tracker(comm);
  const size_t DW = in.D[0], DH = in.D[1], DD = in.D[2];
  // json_interface.hh:634-656
  const bool push_all = getenv("FTK_REF_PUSH_ALL") != NULL && in.nv == 1;
  if (in.nv == 1) {
    tracker.set_scalar_field_source(ftk::SOURCE_GIVEN);
    tracker.set_vector_field_source(push_all ? ftk::SOURCE_GIVEN : ftk::SOURCE_DERIVED);
    tracker.set_jacobian_field_source(push_all ? ftk::SOURCE_GIVEN : ftk::SOURCE_DERIVED);
    tracker.set_jacobian_symmetric(true);
    if (in.nd == 2) tracker.set_domain(ftk::lattice({2, 2}, {DW - 3, DH - 3}));
    else tracker.set_domain(ftk::lattice({2, 2, 2}, {DW - 3, DH - 3, DD - 3}));
  } else {
    tracker.set_scalar_field_source(ftk::SOURCE_NONE);
    tracker.set_vector_field_source(ftk::SOURCE_GIVEN);
    tracker.set_jacobian_field_source(ftk::SOURCE_DERIVED);
    tracker.set_jacobian_symmetric(false);
    if (in.nd == 2) tracker.set_domain(ftk::lattice({1, 1}, {DW - 2, DH - 2}));
    else tracker.set_domain(ftk::lattice({1, 1, 1}, {DW - 2, DH - 2, DD - 2}));
  }
  if (in.nd == 2) tracker.set_array_domain(ftk::lattice({0, 0}, {DW, DH}));
  else tracker.set_array_domain(ftk::lattice({0, 0, 0}, {DW, DH, DD}));
  if (nthreads > 0) tracker.set_number_of_threads(nthreads);
  tracker.set_enable_robust_detection(robust);
  if (type_filter) tracker.set_type_filter(type_filter);
  if (degrees) tracker.set_enable_computing_degrees(true);
  if (!bounds.empty()) tracker.set_coords_bounds(bounds);   // REGULAR_COORDS_BOUNDS, regular_tracker.hh:38
  // FTK_REF_STREAMING: enable_streaming_trajectories (critical_point_tracker.hh:38): trajectories grow after every interval sweep
  // (trace_critical_points_online, critical_point_tracker.hh:523-639), the discrete points are consumed, finalize() traces nothing
  if (getenv("FTK_REF_STREAMING")) tracker.set_enable_streaming_trajectories(true);
  // REGULAR_COORDS_RECTILINEAR / _EXPLICIT (regular_tracker.hh:39-40) with closed-form, exactly representable coordinates:
  //   FTK_REF_COORDS=rect       axis d: 0.5 i + 0.0625 ((i (d + 3)) mod 5) + d
  //   FTK_REF_COORDS=explicit2  E(c, x, y) = 0.75 (c == 0 ? x : y) + 0.03125 ((3 x + 5 y + c) mod 11), two components
  //   FTK_REF_COORDS=explicit3  the same with a third component 0.75 + 0.03125 ((3 x + 5 y + 2) mod 11)
  if (const char *cm = getenv("FTK_REF_COORDS")) {
    const std::string mode(cm);
    if (mode == "rect") {
      std::vector<ftk::ndarray<double>> rc(in.nd);
      for (int d = 0; d < in.nd; d ++) {
        rc[d].reshape((size_t)in.D[d]);
        for (int i = 0; i < in.D[d]; i ++) rc[d][i] = 0.5 * i + 0.0625 * ((i * (d + 3)) % 5) + d;
      }
      tracker.set_coords_rectilinear(rc);
    } else {
      const int nc = mode == "explicit3" ? 3 : 2;
      ftk::ndarray<double> ec;
      ec.reshape((size_t)nc, DW, DH);
      for (size_t y = 0; y < DH; y ++) for (size_t x = 0; x < DW; x ++) for (int c = 0; c < nc; c ++)
        ec(c, x, y) = 0.75 * (c == 0 ? (double)x : c == 1 ? (double)y : 1.0) + 0.03125 * (double)((3 * x + 5 * y + c) % 11);
      tracker.set_coords_explicit(ec);
    }
  }
#ifdef FTKX_SHIM
  tracker.use_accelerator("hip");      // filter.hh (patched): FTK_XL_HIP -- update_timestep() takes the accelerator branch into libftkx.so
  // FTK_SHIM_ONESHOT: the patched tracker without resident snapshots -- every update_timestep() hands host V, J, S to extract_cp*dt_hip
  // (the literal reference boundary); default: snapshots resident in HBM (critical_point_tracker_regular::hip_push_snapshot)
  if (getenv("FTK_SHIM_ONESHOT")) tracker.set_hip_resident(false);
#endif
  tracker.initialize();
  // FTK_REF_T0: the series starts at a later timestep (tracker::set_current_timestep, filters/tracker.hh:40) -- with a large value
  // the int products of element::to_integer (simplicial_regular_mesh.hh:496-502) and the int truncation of simplex_indices
  // (regular_tracker.hh:188-194) wrap on a mesh small enough for the CPU
  const int T0 = getenv("FTK_REF_T0") ? atoi(getenv("FTK_REF_T0")) : 0;
  if (T0) tracker.set_current_timestep(T0);

  // the loop of the reference's callers (json_interface.hh:690-725): push, advance, and a last update for the final ordinal sweep.
  // loop_seconds = everything the tracker does per step (push + update_timestep + pop), without the making of the input array
  double loop_seconds = 0;
  bool resident = false;
  for (int k = 0; k < in.DT; k ++) {
    const auto a = make_array(in, k);
    const auto t0 = std::chrono::high_resolution_clock::now();
    if (push_all) {
      // FTK_REF_PUSH_ALL: scalar, vector and Jacobian all GIVEN (critical_point_tracker::push_field_data_snapshot, critical_point_tracker.hh:202-213),
      // derived here with the reference's own functions -- the arrays the tracker would have derived itself
      ftk::ndarray<double> V, J;
      if (in.nd == 2) { V = ftk::gradient2D(a); J = ftk::jacobian2D<double, true>(V); }
      else { V = ftk::gradient3D(a); J = ftk::jacobian3D(V); }
      tracker.push_field_data_snapshot(a, V, J);
    } else
    if (in.nv == 1) tracker.push_scalar_field_snapshot(a);
    else tracker.push_vector_field_snapshot(a);
    if (k != 0) tracker.advance_timestep();
    if (k == in.DT - 1) tracker.update_timestep();
    resident = resident || tracker.resident();
    loop_seconds += std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
  }

  const auto cps = tracker.get_critical_points();
  fprintf(stdout, "{\"sweep_seconds\": %.6f, \"push_seconds\": %.6f, \"loop_seconds\": %.6f, \"steps\": %d, \"records\": %zu, \"nthreads\": %d, \"hip_resident\": %s}\n",
      tracker.sweep_seconds, tracker.push_seconds, loop_seconds, in.DT, cps.size(), tracker.get_number_of_threads(), resident ? "true" : "false");
  if (getenv("FTK_REF_PER_CALL")) {      // one line more: the milliseconds of every push and every update_timestep()
    fprintf(stdout, "{\"push_ms\": [");
    for (size_t i = 0; i < tracker.push_ms.size(); i ++) fprintf(stdout, "%s%.4f", i ? ", " : "", tracker.push_ms[i]);
    fprintf(stdout, "], \"update_ms\": [");
    for (size_t i = 0; i < tracker.sweep_ms.size(); i ++) fprintf(stdout, "%s%.4f", i ? ", " : "", tracker.sweep_ms[i]);
    fprintf(stdout, "]}\n");
  }
  if (!out) return;

  FILE *fp = fopen(out, "wb");
  if (!fp) { perror(out); exit(2); }
  const char magic[8] = {'F', 'T', 'K', 'R', 'E', 'F', '1', 0};
  fwrite(magic, 1, 8, fp);
  const int32_t h[6] = {in.nd, in.nv, in.D[0], in.D[1], in.D[2], in.DT};
  fwrite(h, sizeof(int32_t), 6, fp);
  const uint64_t nf = tracker.factors.size();
  fwrite(&nf, sizeof(uint64_t), 1, fp);
  for (size_t i = 0; i < nf; i ++) {
    const int64_t step = tracker.factor_steps[i] - T0;   // relative to the first timestep
    fwrite(&step, sizeof(int64_t), 1, fp);
    fwrite(&tracker.factors[i], sizeof(uint64_t), 1, fp);
  }
  const uint64_t nrec = cps.size();
  fwrite(&nrec, sizeof(uint64_t), 1, fp);
  for (const auto &cp : cps) {   // 8 + 4 + 4 + 4 + 4(pad) + 5*8 = 64 bytes
    const uint64_t tag = cp.tag;
    const uint32_t type = cp.type;
    const int32_t ordinal = cp.ordinal, timestep = cp.timestep, pad = 0;
    const double v[5] = {cp.x[0], cp.x[1], cp.x[2], cp.t, cp.scalar[0]};
    fwrite(&tag, 8, 1, fp); fwrite(&type, 4, 1, fp); fwrite(&ordinal, 4, 1, fp);
    fwrite(&timestep, 4, 1, fp); fwrite(&pad, 4, 1, fp); fwrite(v, 8, 5, fp);
  }
  for (int k = 0; k < in.DT; k ++)
    fwrite(in.steps[k].data(), sizeof(double), in.steps[k].size(), fp);
  // the reference's own record-stream writers (filters/critical_point_tracker.hh:108, 339-343, 416-420; json_interface.hh:822-828)
  const char *wprefix = getenv("FTK_REF_WRITE_PREFIX");
  if (wprefix) {
    const std::string p(wprefix);
    tracker.write_critical_points_json(p + ".discrete.json");
    tracker.write_critical_points_binary(p + ".discrete.bin");
    tracker.write_critical_points_text(p + ".discrete.txt");
  }
  // pass 2 of the reference (finalize -> trace_critical_points_offline, filters/critical_point_tracker.hh:668-817):
  // the traced curves, each as the ordered list of its points' element tags
  tracker.finalize();
  const auto &curves = tracker.get_traced_critical_points();
  const char cmagic[4] = {'C', 'U', 'R', 'V'};
  fwrite(cmagic, 1, 4, fp);
  const uint64_t ncurves = curves.size();
  fwrite(&ncurves, 8, 1, fp);
  for (const auto &kv : curves) {
    const int32_t loop = kv.second.loop, n = (int32_t)kv.second.size();
    fwrite(&loop, 4, 1, fp); fwrite(&n, 4, 1, fp);
    for (const auto &cp : kv.second) { const uint64_t tag = cp.tag; fwrite(&tag, 8, 1, fp); }
  }
  // the reference's trajectory post-processing, called exactly as json_interface::post_process does with its default options
  // (filters/json_interface.hh:758-800: no duration pruning, interval points kept, no velocities)
  {
    auto &trajs = tracker.get_traced_critical_points();
    trajs.foreach([](ftk::feature_curve_t &t) { t.smooth_ordinal_types(); t.smooth_interval_types(); t.rotate(); t.update_statistics(); });
    trajs.split_all();
    trajs.foreach([](ftk::feature_curve_t &t) { t.reorder(); t.adjust_time(); t.update_statistics(); });
    if (wprefix) {
      const std::string p(wprefix);
      tracker.write_traced_critical_points_json(p + ".traced.json");
      tracker.write_traced_critical_points_binary(p + ".traced.bin");
      tracker.write_traced_critical_points_text(p + ".traced.txt");
    }
    const char pmagic[4] = {'P', 'P', 'C', 'V'};
    fwrite(pmagic, 1, 4, fp);
    const uint64_t np = trajs.size();
    fwrite(&np, 8, 1, fp);
    for (const auto &kv : trajs) {
      const int32_t loop = kv.second.loop, n = (int32_t)kv.second.size();
      fwrite(&loop, 4, 1, fp); fwrite(&n, 4, 1, fp);
      for (const auto &cp : kv.second) {
        const uint64_t tag = cp.tag; const uint32_t type = cp.type, pad = 0; const double t = cp.t;
        fwrite(&tag, 8, 1, fp); fwrite(&type, 4, 1, fp); fwrite(&pad, 4, 1, fp); fwrite(&t, 8, 1, fp);
      }
    }
  }
  fclose(fp);
}

static void dump_tables(const char *out)
{
  FILE *fp = fopen(out, "w");
  for (int n = 3; n <= 4; n ++) {
    ftk::simplicial_regular_mesh m(n);
    const int d = n - 1;
    fprintf(fp, "mesh %d dim %d ntypes %d ordinal %d interval %d\n", n, d,
        m.ntypes(d), m.ntypes(d, ftk::ELEMENT_SCOPE_ORDINAL), m.ntypes(d, ftk::ELEMENT_SCOPE_INTERVAL));
    for (int t = 0; t < m.ntypes(d); t ++) {
      ftk::simplicial_regular_mesh_element e(std::vector<int>(n, 0), d, t);
      fprintf(fp, "type %d ordinal %d :", t, (int)e.is_ordinal(m));
      for (const auto &v : e.vertices(m)) {
        fprintf(fp, " ");
        for (int c : v) fprintf(fp, "%d", c);
      }
      fprintf(fp, " | side_of:");
      for (const auto &c : e.side_of(m)) {
        fprintf(fp, " (%d;", c.type);
        for (int x : c.corner) fprintf(fp, " %d", x);
        fprintf(fp, ")");
      }
      fprintf(fp, "\n");
    }
    // sides of every (d+1)-cell type
    for (int t = 0; t < m.ntypes(d + 1); t ++) {
      ftk::simplicial_regular_mesh_element e(std::vector<int>(n, 0), d + 1, t);
      fprintf(fp, "cell %d sides:", t);
      for (const auto &s : e.sides(m)) {
        fprintf(fp, " (%d;", s.type);
        for (int x : s.corner) fprintf(fp, " %d", x);
        fprintf(fp, ")");
      }
      fprintf(fp, "\n");
    }
  }
  fclose(fp);
}

int main(int argc, char **argv)
{
  if (argc < 2) { fprintf(stderr, "see header comment for usage\n"); return 2; }
  const std::string mode = argv[1];
  const bool robust = getenv("FTK_REF_NO_ROBUST") == NULL;
  const unsigned type_filter = getenv("FTK_REF_TYPE_FILTER") ? atoi(getenv("FTK_REF_TYPE_FILTER")) : 0;
  const bool degrees = getenv("FTK_REF_DEGREES") != NULL;
  std::vector<double> bounds;
  if (const char *b = getenv("FTK_REF_BOUNDS")) {   // "x0,x1,y0,y1[,z0,z1]"
    std::string sb(b); size_t pos = 0;
    while (pos < sb.size()) { size_t q = sb.find(',', pos); if (q == std::string::npos) q = sb.size(); bounds.push_back(atof(sb.substr(pos, q - pos).c_str())); pos = q + 1; }
  }
  if (mode == "tables") { dump_tables(argv[2]); return 0; }

  input_t in;
  const char *out = NULL;
  int nthreads = 0;
  if (mode == "synthetic") {
    if (argc < 8) return 2;
    double x0dir[6] = {10, 10, 10, 0.1, 0.11, 0.1};
    const std::string name = argv[2];
    if (name == "moving_extremum_2d") { x0dir[3] = 0.1; x0dir[4] = 0.1; }
    out = argv[7];
    int a = 8;
    if (argc >= 14) { for (int i = 0; i < 6; i ++) x0dir[i] = atof(argv[8 + i]); a = 14; }
    if (argc > a) nthreads = atoi(argv[a]);
    in = synthesize(name, atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), atoi(argv[6]), x0dir);
  } else if (mode == "file") {
    in = read_input(argv[2]);
    out = argv[3];
    if (argc > 4) nthreads = atoi(argv[4]);
  } else if (mode == "time") {
    in = read_input(argv[2]);
    if (argc > 3) nthreads = atoi(argv[3]);
  } else return 2;

  if (in.nd == 2) run<ftk::critical_point_tracker_2d_regular>(in, nthreads, out, robust, type_filter, degrees, bounds);
  else run<ftk::critical_point_tracker_3d_regular>(in, nthreads, out, robust, type_filter, degrees, bounds);
  return 0;
}
