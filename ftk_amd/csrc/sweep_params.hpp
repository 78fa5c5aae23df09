// Launch parameters shared by the host API and the HIP kernels.
#pragma once
#include "../../include/ftkx.h"

namespace ftkx {

typedef unsigned long long u64;
typedef long long i64;

enum { CNT_HITS = 0, CNT_CELLS_SURVIVED = 1, CNT_SIMPLICES_TESTED = 2, CNT_SURVIVOR_LIST = 3, CNT_LIST_PEAK = 4,
       CNT_REFINE_LIST = 5, CNT_REFINE_PEAK = 6, CNT_WORDS_REFINED = 7, CNT_PASS = 8, CNT_SPARSE = 9, CNT_FRAGILE = 10,
       CNT_SERIES_DONE = 11,   // series pass: non-zero once the pass has been finished early (sparse data: one workgroup did the whole tail) --
                               // the kernels queued behind that point leave at once
       CNT_BUCKET_MAX = 12,    // series pass: the fullest bucket of the ordering step
       CNT_SMALL_DONE = 13,    // series pass: workgroups of the fused tail kernel that have finished (32-bit counter in this word)
       CNT_N = 14 };

// A simplex that passed the test, handed from the integer kernels (exact_kernel, tile_kernel) to record_kernel, which does all the
// FP64 work (solve, lerp, Jacobian, classification) on densely packed lanes: corner index inside core (x fastest) | type | request.
constexpr int kPassTypeShift = 40, kPassStepShift = 46;
inline constexpr u64 kPassLinMask = (1ull << kPassTypeShift) - 1ull;

// everything that does not change between the sweeps of one context configuration
struct Mesh {
  int nd;
  int dom_lb[3], dom_ub[3];  // inclusive vertex validity box (mesh lb/ub, spatial part)
  int core_st[3], core_sz[3];
  int ext_st[3], ext_sz[3];
  u64 mesh_prod[4];          // lattice::prod_ of the mesh lattice -> SoS vertex ids (regular_tracker.hh:188-194)
  int dimprod[4];            // simplicial_regular_mesh::dimprod_ (int) -> reference tag
  u64 exact_prod[4];         // same in 64 bits
  int mask_pitch;            // row pitch (bytes) of the vertex-mask arrays: roundup8(ext_sz[0]) + 8
  int u_pitch;               // row pitch of the per-8-vertex summary arrays: roundup8(ceil(ext_sz[0] / 8)) + 8
  int u_rows;                // rows a summary byte stands for: 1 (a word of 8 vertices) or 4 (the 8 x 4 block of aligned rows; row
                             // index of U = y / u_rows, ceil(ext_sz[1] / u_rows) rows per plane) -- mask_summary_rows()
  int jacobian_symmetric, robust, use_type_filter;
  unsigned type_filter;
  int compute_degrees, tag_mode;
  int scalar_mode;           // 1: V is not stored; it is gradient2D/3D(S) evaluated where needed (vector_field_source == DERIVED)
  int derive_jacobian;       // 1: J not stored; jacobian2D/3D of V evaluated at hit vertices (jacobian_field_source == DERIVED)
  int record_general;        // 1: every record takes the general gather (FTKX_RECORD_GENERAL=1: cross-check of the straight-line one)
  int coords_mode;           // REGULAR_COORDS_*: 0 lattice integers, 1 image bounds, 2 rectilinear, 3 explicit
  double coords_bounds[6];
  const double *coords_rect[3];   // RECTILINEAR: per-axis coordinate arrays (device), indexed by the vertex coordinate
  const double *coords_expl;      // EXPLICIT: (ncomp, n0, ...) array (device), read as p[c + ncomp * (x + n0 * y)]
  int coords_expl_ncomp, coords_expl_n0;
  ftkx_cp_t *hits;           // device hit buffer
  u64 *pass;                 // simplices that passed the test, awaiting record_kernel (same capacity as hits)
  u64 *counters;             // CNT_* device counters
  u64 capacity;              // records the hit buffer can hold
  u64 *fragile;              // 3D records whose class hangs on the last bits of libm (classify3): 10 words each -- slot in `hits`, J[3][3]
  u64 fragile_capacity;
  // series pass (series.hip): the exact kernel also counts the simplices that passed per bucket of their order key
  unsigned *hist;            // nullptr: no histogram
  int hist_shift;            // bucket = order_key >> hist_shift
  u64 core_cells;            // corners in core: order_key = ((step * core_cells + corner index) << 6) | type
};

// The order of the records of one series pass: by (step, corner index inside core, simplex type) -- which IS the order of the element
// tags when the steps come in ascending time and the tags do not wrap (simplicial_regular_mesh.hh:496-502).  A pass descriptor
// (kPass* below) rearranged into one ascending integer:
__host__ __device__ inline u64 order_key(u64 pass_desc, u64 core_cells)
{
  const u64 lin = pass_desc & ((1ull << 40) - 1ull), type = (pass_desc >> 40) & 63ull, step = pass_desc >> 46;
  return ((step * core_cells + lin) << 6) | type;
}

// the fields of one (timestep, scope) request: slice t and slice t+1
struct Fields {
  const double *S[2];
  const double *V[2];
  const double *J[2];
  const unsigned char *M[2]; // vertex sign masks (fast path only)
  const unsigned char *U[2]; // their per-word summaries (two-level cull), or nullptr
  double factor;             // (double)vector_field_scaling_factor, a power of two
  int t;                     // current_timestep
  int scope_mask;            // FTKX_SCOPE_*
};

constexpr int kMaskKernels = 7;      // mask-kernel families launch_masks picks from (mask_kernels.hip: mask_kernel_launches)

// tile kernel (exact_only / non-robust / overflow regime, and small jobs)
struct TileParams {
  Mesh m;
  const Fields *steps;       // the requests of this launch in the batch's device array of Fields, in time order: a workgroup keeps its tile
  int nsteps;                // for all of them, and a slice two consecutive steps share is staged once (tile_kernels.hip)
  int cull;                  // 1: strict-sign cull legal and enabled
  int repeat;                // > 1: the fan phase alone that many times on the staged tile (ftkx_debug_tile_repeat); records are not affected
  int ntiles[3];
  int step;                  // index of steps[0] in the batch's device array of Fields (what record_kernel looks a simplex's request up by)
  int fan;                   // the tests' knob (FTKX_TILE_FAN): 0 = (corner, type) pairs over the lanes everywhere, 1 = no fp64 fan, 2 = as the tiles allow
  int form;                  // which tile_kernel<ND, FORM> to launch: from the slices' largest magnitude x factor where the context knows it
  u64 *stats;                // 256 x {simplices tested, cells survived}: a workgroup adds to slot blockIdx % 256, tile_stats_fold_kernel sums
                             // them into the counters (one address for 65 536 workgroups is 0.65 ms of serialised atomics per launch)
};

// Largest M with 24 M^3 < 2^63 (3D: |det4| of a homogeneous 4x4 with entries <= M) / 6 M^2 < 2^63 (2D): a simplex all of whose
// quantised components satisfy |q| + 1 <= M cannot wrap any determinant of the predicate, so the strict-sign cull is exact for it.
constexpr i64 kSafeM3 = 727041, kSafeM2 = 1239850262;
template <int ND> constexpr i64 safe_m() { return ND == 3 ? kSafeM3 : kSafeM2; }

// mask kernel job: one slice.
// The masks are built under a quantisation factor F = 1 / threshold that may be SMALLER than the factor a sweep later uses:
// v >= 1/F implies trunc(v * F') >= 1 for every F' >= F, so such masks only ever cull less (the factor is a sticky running
// minimum of resolutions, i.e. it only grows: the masks of a slice can be built before its own reduction is known).
// `big` = safe_m / F: a vertex with some |v| >= big could take a determinant out of int64 under F; it gets NO sign bits and
// therefore never takes part in a cull.  Under a larger factor more vertices are big: the host reuses masks built under F for
// F' > F only when the slice's max |v| shows that no vertex is big under F' (ftkx_api.hip, masks_valid).
struct MaskJob {
  const double *S;
  const double *V;
  unsigned char *M;
  unsigned char *U;          // summary: AND of the 8 mask bytes of each aligned x word (nullptr: not produced)
  u64 *red;                  // reduction output, 64 slots of {min, max} as IEEE bit patterns (nullptr: none).
                             //   pre-pass instantiations (REDUCE): min over ALL non-zero finite |v| = ndarray::resolution();
                             //   fused into the mask kernels: min over the non-zero |v| < threshold only (all the scaling
                             //   factor needs: anything >= threshold = 1/F cannot push nbits past log2 F), DBL_MAX if none.
                             //   max: largest |v|, +Inf if the slice holds an Inf.
  double threshold;          // 1 / F: q = trunc(v * F) > 0  <=>  v >= 1/F (F is a power of two)
  double big;                // safe_m / F  (+Inf: no vertex is ever big)
  double tx, ty;             // 2D scalar input: the smallest T with fl(T (D - 1)) >= threshold per axis -- gradient2D's scaling folded into the
                             // compare (set_lean_thresholds; 0: not set, the kernel finds them itself)
};

// The smallest T with fl(T * f) >= thr (f an integer-valued double >= 1, thr > 0): rounding is monotone, so fl(d * f) >= thr <=> d >= T and,
// round-to-nearest being symmetric, fl(d * f) <= -thr <=> d <= -T, for every d including the infinities; a NaN compares false both ways with or
// without the multiplication.  A division and at most a few steps to a neighbouring double.  f == 0 (a slice one row high: its y differences
// are x - x = 0 or NaN) gives +Inf: no difference passes, as no product with 0 does.
__host__ __device__ inline double exact_threshold(double thr, double f)
{
  double T = thr / f;
  if (!(T < 1.7976931348623157e308)) return __builtin_huge_val();
  for (int it = 0; it < 4 && T * f < thr; it ++) { long long b; __builtin_memcpy(&b, &T, 8); b ++; __builtin_memcpy(&T, &b, 8); }
  for (int it = 0; it < 4; it ++) {
    long long b; __builtin_memcpy(&b, &T, 8); b --;
    double below; __builtin_memcpy(&below, &b, 8);
    if (below > 0.0 && below * f >= thr) T = below; else break;
  }
  return T;
}

constexpr int kSeriesMaxSlices = 2048;    // slices (and steps) one series pass takes: their reductions are folded in LDS
constexpr int kSeriesMaxBins = 1 << 16;    // buckets of the ordering step
constexpr int kFoldMaxSlices = 256;        // up to this many slices the factor job rides in the cull kernel (FactorJob); above: its own kernel

// ---- series pass: sticky factor on the device (critical_point_tracker.hh:850-864) -----------------------------------------------
struct SeriesSlice {
  int t;
  int red_index;             // which 64-slot block of `red` holds this slice's fused reduction; -1: reduced earlier (known_* stand)
  double known_res;          // smallest non-zero |v| below 1 / (the factor its masks were built under), or DBL_MAX; DBL_MAX if not known
  double known_max;          // max |v|; 0 if not known
  const u64 *from_res, *from_max;   // non-null: the two values as the pass queued before this one leaves them in ITS results block (the host has not seen them yet)
};
struct SeriesStep;
struct SeriesSlice;
struct Fields;
// The arguments of series_factors_body when the job is done by one extra workgroup of the cull kernel (which does not need the factors,
// only what the mask kernel left: the two run side by side instead of one behind the other -- a kernel boundary and a one-workgroup
// kernel less per pass).  enabled = 0: no such workgroup.
struct FactorJob {
  Fields *steps; const SeriesSlice *slices; const SeriesStep *sinfo; const u64 *red; const u64 *running_from; u64 *results; u64 *counters;
  double running_in, safe_m;
  int nsteps, nslices, enabled, pad;
};
struct SeriesStep {
  int slice0, slice1;        // indices of the step's slices (slice1 = -1: ordinal sweep only)
  int last;                  // the sticky minimum of this step runs over slices 0 .. last (every slice with timestep <= t + 1)
  int pad;
};
enum { SERIES_AMBIGUOUS = 1,        // 1 / resolution so close above a power of two that the last bit of the host's log2 decides nbits
       SERIES_MASKS_INVALID = 2,    // a slice has vertices that could overflow a determinant under its step's factor: masks need the per-vertex rule
       SERIES_INF = 4,              // a slice holds an Inf: the fused maximum is not the max FINITE |v|
       SERIES_OVERFLOW = 8,         // a list / pass / fragile buffer was too small
       SERIES_FIX_ORDER = 16,       // a bucket of the ordering step was too full to rank on the device: its records are unordered among themselves
       SERIES_EARLY = 32,           // (informational) the fused tail kernel finished the pass
       SERIES_LATE_DECLINE = 64,    // the fused tail found more records than its last workgroup orders, took back what it had counted and left the pass to
                                    // the bucket chain (informational: the records are complete and ordered; rounds 3-4a sent them unordered for the host to sort)
       SERIES_TAIL_PENDING = 128,   // the pass was queued in its short form (mask, cull, fused tail) and the fused tail declined: the host queues the rest
       SERIES_ONE = 512,            // (informational) the one-launch pass for small series did it (one_kernel.hip)
       SERIES_HALO_FULL = 256 };    // slab pass (ftkx_series_dist_*): the halo slice is needed as a whole (too many surviving cells for a request, a mask
                                    // message that did not fit, masks the host will rebuild): the request said -1, nothing was swept
// results block (device copy and coherent pinned copy, same layout; u64 words)
enum { SR_STATUS = 0, SR_RUNNING = 1, SR_NHITS = 2, SR_NFRAGILE = 3, SR_SPARE = 4,
       SR_HALO_ASKED = 5,    // slab pass: cells this rank asked its upper neighbour for (-1: the whole slice), 0 without a halo
       SR_HALO_SERVED = 6,   // slab pass: what the lower neighbour asked this rank for (-1: it needs this rank's first slice as a whole)
       SR_COUNTERS = 7, SR_HEAD = 7 + CNT_N };   // then factors[nsteps], res[nslices], max[nslices], [slab pass: 4 words per rank as gathered], fragile[cap * 10]
// slab pass: the per-pass block of device words next to the results (series.hip): counters of the halo hand-over and a stub shaped like a
// results block whose SR_RUNNING word is the running minimum BEFORE this rank's slab (what FactorJob::running_from reads)
enum { DB_WORDS = 0 /* mask words compacted into the outgoing message */, DB_BAD = 1 /* the incoming mask message did not fit / did not match */,
       DB_DONE = 2 /* workgroups of the export kernel that have finished */, DB_BAD2 = 3 /* the incoming message named a word outside the mask array */, DB_PSEUDO = 8, DB_N = 16 };
// the one-launch pass for small series (one_kernel.hip): everything the kernel needs comes by value, in its arguments
constexpr int kOneMaxSlices = 48, kOneMaxSteps = 47, kOneParts = 32;      // (parts: at most; a slice's reduction is cut into min(32, ceil(workgroups / slices)) chunks)
constexpr unsigned kOneKeys = 1024;             // order keys a workgroup parks in LDS
struct OneSlice { const double *S, *V, *J; int t, pad; };
struct OneStep { int t, scope, slice0, slice1, last, pad; };
struct OneArgs {
  int nsteps, nslices;
  OneSlice slice[kOneMaxSlices];
  OneStep step[kOneMaxSteps];
  double running_in, cap;                      // the running minimum before this pass; 1 / hint: only components below it can lower the minimum
  const u64 *running_from;                     // a results block on the device whose SR_RUNNING word this pass continues from (nullable)
  u64 *scratch;                                // ONE_* words, zero between launches
  u64 *results, *h_results; size_t nwords; unsigned *flag; unsigned seq;
  ftkx_cp_t *out; u64 capacity; u64 *fragile; u64 fragile_capacity;
};
constexpr unsigned kOneMaxBlocks = 8192;        // blocks of (step, corner) a pass is cut into (a block = one staging batch of the kernel), dealt to the workgroups round-robin
constexpr unsigned kOneOwnBlocks = 64;          // ... of which a workgroup takes at most this many
enum { ONE_BAR = 0 /* four 32-bit words: two barriers, arrivals at the exit, "somebody gave up" */, ONE_TESTED = 2, ONE_NFRAG = 3, ONE_CELLS = 4, ONE_OVER = 5,
       ONE_BCOUNT = 8 /* one 32-bit count per block */, ONE_PARTS = 8 + kOneMaxBlocks / 2, ONE_WORDS = 8 + kOneMaxBlocks / 2 + 2 * kOneMaxSlices * kOneParts };
constexpr int kDistContrib = 4;            // words a rank contributes to the all_gather: slab min resolution, slab max |v|, the same of its FIRST slice

}  // namespace ftkx
