/* TEST INFRASTRUCTURE -- NOT PART OF THE PRODUCT.
 *
 * Plain-C, CPU-only restatement of the reference (hguo/ftk) critical-point space-time simplex
 * sweep.  It exists to CHECK the HIP path in ftk_amd/: only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.  Nothing under ftk_amd/ links, imports or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this restatement, record for record
 * (tag/type/ordinal/timestep exact, coordinates and scalar bit-identical), against fixtures in
 * tests/golden/ that were produced by the real reference CPU path compiled from /root/reference
 * (oracle/ref_driver.cpp, oracle/Makefile), plus the known-answer counts of BASELINE.md section 3.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference/include/ftk/ unless noted).
 */
#ifndef FTK_ORACLE_H
#define FTK_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* == ftk::feature_point_lite_t, features/feature_point_lite.hh:8-15 (72 bytes) */
typedef struct {
  double x[3];
  double t;
  double scalar[3];
  unsigned int type;
  unsigned long long tag;
} ftko_cp_t;

/* one emitted record plus the element it came from */
typedef struct {
  ftko_cp_t cp;
  int ordinal;     /* e.is_ordinal(m)                        */
  int timestep;    /* current_timestep of the emitting sweep */
  int etype;       /* simplex type id in the ALL-scope table */
  int pad;
  int corner[4];   /* x, y, [z,] t of the element's corner   */
} ftko_rec_t;

enum { FTKO_SCOPE_ORDINAL = 1, FTKO_SCOPE_INTERVAL = 2 };            /* mesh/simplicial_regular_mesh.hh:39-43 */
enum { FTKO_TAG_WORK_INDEX = 0, FTKO_TAG_REFERENCE = 1, FTKO_TAG_EXACT64 = 2 };

/* One sweep of one scope at one timestep == one call of
 * regular_tracker::element_for(ordinal, k, func) (filters/regular_tracker.hh:196-211), equivalently one
 * call across the accelerator boundary extract_cp{2,3}dt_* (filters/critical_point_tracker_2d_regular.hh:33-63,
 * critical_point_tracker_3d_regular.hh:42-56) but with the CPU path's semantics. */
typedef struct {
  int nd;                                 /* 2 or 3 spatial dimensions */
  int scope;                              /* FTKO_SCOPE_*              */
  int current_timestep;
  long long domain_st[3], domain_sz[3];   /* vertex validity box == mesh lb/ub (spatial part); time is [0, INT_MAX] */
  long long core_st[3],   core_sz[3];     /* corners to enumerate (local_domain), x fastest                        */
  long long ext_st[3],    ext_sz[3];      /* array lattice (local_array_domain) used to index V/J/S               */
  const double *V[2];                     /* [nd, ext] current / next; V[1] may be NULL for ordinal                */
  const double *J[2];                     /* [nd, nd, ext] or NULL                                                  */
  const double *S[2];                     /* [ext] or NULL (== scalar_field_source NONE)                            */
  unsigned long long factor;              /* vector_field_scaling_factor                                            */
  int jacobian_symmetric;                 /* is_jacobian_field_symmetric                                            */
  int robust;                             /* enable_robust_detection (3D only; 2D ignores it)                       */
  int use_type_filter;                    /* 2D only (filters/critical_point_tracker_2d_regular.hh:280)             */
  unsigned int type_filter;
  int compute_degrees;                    /* enable_computing_degrees (2D only)                                     */
  int tag_mode;                           /* FTKO_TAG_*                                                             */
  int nthreads;                           /* <=1: serial                                                            */
  int coords_mode;                        /* 0 REGULAR_COORDS_SIMPLE, 1 _BOUNDS, 2 _RECTILINEAR, 3 _EXPLICIT (regular_tracker.hh:12-17, 38-40;
                                             simplex_coordinates 2d:494-527, 3d:342-378)                            */
  double bounds[6];                       /* x0,x1,y0,y1[,z0,z1]                                                    */
  const double *rect[3];                  /* RECTILINEAR: one array per axis, indexed by the vertex coordinate      */
  const double *expl;                     /* EXPLICIT: ndarray (expl_ncomp, expl_n0, ...), read as p[c + ncomp*(x + n0*y)] */
  int expl_ncomp, expl_n0;
} ftko_sweep_args;

/* returns the number of records, writes a malloc'd array sorted by (corner t,z,y,x, type) to *out (free with ftko_free) */
size_t ftko_sweep(const ftko_sweep_args *a, ftko_rec_t **out);
void ftko_free(void *p);

/* number of work items of one sweep: core.n() * ntypes(d, scope) (mesh/simplicial_regular_mesh.hh:1042) */
unsigned long long ftko_num_work_items(const ftko_sweep_args *a);

/* ---- the simplex fan (mesh/simplicial_regular_mesh.hh:620-715, 799-831) ---- */
/* n = mesh dimension (3: 2D+t, 4: 3D+t).  Returns ntypes of (n-1)-simplices; fills
 * verts[type][vertex][axis] (0/1), is_ordinal[type]. */
int ftko_unit_simplices(int n, int verts[60][4][4], int is_ordinal[60]);
/* side_of relation of an (n-1)-simplex type: the two n-cells (type, corner offset) that contain it
 * (mesh/simplicial_regular_mesh.hh:717-797).  Returns count (always 2). */
int ftko_side_of(int n, int type, int cell_type[2], int cell_offset[2][4]);
/* sides of an n-cell type: n+1 faces (type, corner offset) */
int ftko_sides(int n, int cell_type, int face_type[5], int face_offset[5][4]);

/* ---- derived fields (ndarray/grad.hh) and the quantisation factor ---- */
void ftko_gradient2D(const double *S, int DW, int DH, double *V);                       /* grad.hh:10-31   */
void ftko_jacobian2D(const double *V, int DW, int DH, int symmetric, double *J);        /* grad.hh:54-86   */
void ftko_gradient3D(const double *S, int DW, int DH, int DD, double *V);               /* grad.hh:130-149 */
void ftko_jacobian3D(const double *V, int DW, int DH, int DD, double *J);               /* grad.hh:175-212 */
double ftko_resolution(const double *p, size_t n);                                      /* ndarray.hh:770-778 */
unsigned long long ftko_scaling_factor(double resolution, int *nbits);                  /* filters/critical_point_tracker.hh:850-864 */

/* ---- synthetic inputs (ndarray/synthetic.hh, ndarray/stream.hh) ---- */
void ftko_synthetic_woven_2D(int DW, int DH, double t, double *S);                      /* synthetic.hh:29-45 */
void ftko_synthetic_merger_2D(int DW, int DH, double t, double *S);                     /* synthetic.hh:263-298 */
void ftko_synthetic_moving_extremum(int nd, const int *D, const double *x0, const double *dir, double t, double *S); /* synthetic.hh:332-354 */
void ftko_synthetic_double_gyre(int DW, int DH, double time, double A, double omega, double eps, double *V);      /* synthetic.hh:130-150,193-217 */

/* ---- the whole tracker loop for a time series resident in host memory ----
 * Mirrors json_interface::consume_regular (filters/json_interface.hh:606-725) driving
 * critical_point_tracker_{2d,3d}_regular: scalar input (nv == 1) derives V and J and sweeps
 * domain [2, D-2]; vector input (nv == nd) derives J and sweeps [1, D-2]. */
typedef struct {
  int nd, nv;
  int D[3];
  int DT;
  const double *const *steps;   /* DT pointers to nv*D0*D1*D2 doubles (component fastest) */
  int robust, use_type_filter;
  unsigned int type_filter;
  int compute_degrees;
  int tag_mode;
  int nthreads;
  int coords_mode;
  double bounds[6];
  const double *rect[3];
  const double *expl;
  int expl_ncomp, expl_n0;
  int t0;                       /* tracker::set_current_timestep (filters/tracker.hh:40): timestep of steps[0] */
} ftko_track_args;

/* returns #records; *out sorted by (t,z,y,x,type); factors[k] (k < DT) = factor in force at the sweep of
 * current_timestep == k; sweep_seconds (nullable) accumulates the time spent inside the sweeps only. */
size_t ftko_track(const ftko_track_args *a, ftko_rec_t **out, unsigned long long *factors, double *sweep_seconds);

#ifdef __cplusplus
}
#endif
#endif
