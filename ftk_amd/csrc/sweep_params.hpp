// Launch parameters shared by the host API and the HIP kernels.
#pragma once
#include "../../include/ftkx.h"

namespace ftkx {

typedef unsigned long long u64;
typedef long long i64;

enum { CNT_HITS = 0, CNT_CELLS_SURVIVED = 1, CNT_SIMPLICES_TESTED = 2, CNT_SLOW_PATH = 3, CNT_N = 8 };

struct SweepParams {
  int nd;
  int scope_mask;            // FTKX_SCOPE_*
  int t;                     // current_timestep
  int dom_lb[3], dom_ub[3];  // inclusive vertex validity box
  int core_st[3], core_sz[3];
  int ext_st[3], ext_sz[3];
  const double *V[2];        // device pointers: slice t, slice t+1
  const double *J[2];
  const double *S[2];
  double factor;             // (double)vector_field_scaling_factor, a power of two
  u64 mesh_prod[4];          // lattice::prod_ of the mesh lattice -> SoS vertex ids (regular_tracker.hh:188-194)
  int dimprod[4];            // simplicial_regular_mesh::dimprod_ (int) -> reference tag
  u64 exact_prod[4];         // same in 64 bits
  int jacobian_symmetric, robust, use_type_filter;
  unsigned type_filter;
  int compute_degrees, tag_mode;
  int cull;                  // 1: strict-sign cull is legal (robust test, no int64 overflow possible) and enabled
  int derive_jacobian;       // 0: J given; 1: derive from V like jacobian2D<symmetric?>/jacobian3D; see jac_symmetric_derive
  int jac_symmetric_derive;  // jacobian2D<T, true> (scalar input) vs <T, false> (vector input)
  ftkx_cp_t *hits;           // device hit buffer
  u64 *counters;             // CNT_* device counters
  u64 capacity;              // records the hit buffer can hold
  int ntiles[3];
};

}  // namespace ftkx
