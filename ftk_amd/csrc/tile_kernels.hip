// The TILE PATH of the critical-point sweep (gfx950): every simplex of every corner takes the integer test -- `exact_only`, non-robust 3D,
// factors that are no power of two, and the regime where determinants may wrap and most cells survive the cull.  One workgroup per tile of
// corners, the tile's vertex block staged in LDS.  Reference arithmetic: check_simplex (critical_point_tracker_3d_regular.hh:453-464,
// ..._2d_regular.hh:605-622), numeric/sign_det.hh:92-200, 360-414, numeric/det.hh:16-55.  (Split from sweep_kernels.hip in round 6.)
#include "internal.hpp"
#include "sweep_device.hpp"
#include "series_device.hpp"

namespace ftkx {



// ---------------------------------------------------------------------------------------------------------------
// TILE PATH
// ---------------------------------------------------------------------------------------------------------------
template <class F, int... I> __device__ __forceinline__ void fan_for_impl(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F> __device__ __forceinline__ void fan_for(F &&f) { fan_for_impl(f, std::make_integer_sequence<int, N>{}); }

#ifdef FTKX_TILE_STAMPS
__device__ unsigned long long g_tile_stamps[512 * 8];
#define TILE_STAMP(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); phase_[k] = now_ - stamp_; stamp_ = now_; } while (0)
#else
#define TILE_STAMP(k) do { } while (0)
#endif
#ifndef FTKX_FAN_WAVES
#define FTKX_FAN_WAVES 3
#endif

struct fan_result { unsigned hits[2], unsure[2], tested; };     // bit T of the 64: simplex type T

// sum_k p_k * b_k in Z / 2^64 for 64-bit p_k and sign-extended 32-bit b_k: per term ONE 32 x 32 -> 64 multiply-add on the low words,
// one 32-bit multiply for the high word, and the sign of b_k as a mask instead of a third multiply
// (p * b = lo * bu + 2^32 * (hi * bu - [b < 0] * lo), bu = b as unsigned)
__device__ __forceinline__ u64 dot3_64_s32(u64 p0, u64 p1, u64 p2, int b0, int b1, int b2)
{
  const unsigned l0 = (unsigned)p0, l1 = (unsigned)p1, l2 = (unsigned)p2, h0 = (unsigned)(p0 >> 32), h1 = (unsigned)(p1 >> 32), h2 = (unsigned)(p2 >> 32);
  const u64 r = (u64)l0 * (unsigned)b0 + (u64)l1 * (unsigned)b1 + (u64)l2 * (unsigned)b2;
  const unsigned h = (unsigned)(r >> 32) + h0 * (unsigned)b0 + h1 * (unsigned)b1 + h2 * (unsigned)b2
                   - (l0 & (unsigned)(b0 >> 31)) - (l1 & (unsigned)(b1 >> 31)) - (l2 & (unsigned)(b2 >> 31));
  return (u64)(unsigned)r | ((u64)h << 32);
}

// The whole 3D+t fan of ONE corner on one lane.  The generic loop of tile_kernel spreads (corner, type) pairs over the lanes and computes
// every simplex from scratch: twelve LDS reads, a decode of the pair, four 3 x 3 minors.  Here a lane walks the 60 simplex types with
// compile-time vertex indices (`at(v, c)`: component c of hypercube vertex v, an LDS read at a constant offset from the lane's corner).
// A simplex is a chain 0 < m1 < m2 < m3 of vertex masks; three of its four "vertex replaced by the origin" determinants contain the corner,
//   n1 = det(X_0, X_m2, X_m3),  n2 = det(X_0, X_m1, X_m3),  n3 = det(X_0, X_m1, X_m2),
// and are shared between simplices; the fourth, det(X_m1, X_m2, X_m3), is its own.  Invalid / non-finite vertices and the strict-sign cull
// are sixteen-bit masks tested against the simplex's vertex set.  Two arithmetics for the same values:
//   * integer (comp_t = int): |component| < 2^31 (tile_kernel checks).  2 x 2 minors from 32 x 32 -> 64 multiplies, determinants by
//     dot3_64_s32, everything in Z / 2^64 exactly as origin_in_simplex3 -- wrapped determinants included;
//   * fp64 (comp_t = double): |component| < 2^19.  Nothing wraps there (|3 x 3 determinant| < 6 * 2^57, |D| < 42 * 2^57 < 2^63), so the
//     reference's wrapped signs are the integers' true signs, and a double-precision evaluation decides them whenever the value is
//     clear of its rounding error: minors are exact (< 2^39), a determinant is off by less than 2^9, D by less than 2^11; `clear` =
//     4096 then, and 0.5 where |component| < 2^16, since then every intermediate is an integer below 2^53 and the evaluation is
//     exact.  v_fma_f64 is a full-rate instruction on gfx950, a 32 x 32 -> 64 integer multiply a quarter-rate one.
// A simplex with a value that is zero / INT64_MIN (fp64: not clear of zero) comes back in `unsure`: tile_kernel gives it the integer
// test with the literal cascade.
//
// Round 6: the shared determinants by GROUPS of the chain's middle vertex (fan_tables.hpp, fan_groups).  Rounds 3-5 computed all fifty
// det(X_0, X_a, X_b) up front -- a hundred registers that held the kernel at two wavefronts per SIMD.  With c(m) = X_0 x X_m,
//   n1 =  X_m3 . c(m2),   n3 = -X_m1 . c(m2),   n2 = -X_m1 . c(m3):
// per group one cross product c(m2), the n3 of its (at most six) m1, and per m3 (at most three) its n1 and c(m3) -- eighteen doubles live
// at a time, at the price of more arithmetic (23 cross and 118 dot products instead of 15 and 50: +12 % of the fan's instructions) -- and a
// scheduling barrier behind every group, without which the compiler interleaves the groups for latency and is back at 256 registers (or,
// held to 168, at 316 bytes of spills: 28.5 ms).  168 registers, no scratch, three wavefronts per SIMD: 256^3 x 16 exact-only 25.7 ->
// 22.1 ms (a barrier behind every m3 instead: 22.9).  Same values up to rounding in the fp64 form (the error bounds above hold for any
// order of the three products of a determinant), identical ones in the integer form (Z / 2^64).
template <class comp_t, class At>
__device__ __forceinline__ fan_result fan_of_corner3(At &&at, unsigned inv, const unsigned (&pos)[3], const unsigned (&neg)[3],
                                                             bool do_ord, bool do_int, int cull, double clear)
{
  constexpr bool FP = std::is_same<comp_t, double>::value;
  using minor_t = std::conditional_t<FP, double, u64>;
  auto minor = [](comp_t a, comp_t b, comp_t c, comp_t d) __attribute__((always_inline)) -> minor_t {      // a * b - c * d
    if constexpr (FP) return fma(a, b, -(c * d));
    else return (u64)((i64)a * (i64)b) - (u64)((i64)c * (i64)d);
  };
  struct vec3 { minor_t c0, c1, c2; };
  const comp_t o0 = at(std::integral_constant<int, 0>{}, 0), o1 = at(std::integral_constant<int, 0>{}, 1), o2 = at(std::integral_constant<int, 0>{}, 2);
  auto cross0 = [&](auto J) __attribute__((always_inline)) -> vec3 {          // X_0 x X_J
    const comp_t j0 = at(J, 0), j1 = at(J, 1), j2 = at(J, 2);
    return vec3{minor(o1, j2, o2, j1), minor(o2, j0, o0, j2), minor(o0, j1, o1, j0)};
  };
  auto dot = [&](const vec3 &c, auto K) __attribute__((always_inline)) -> minor_t {      // X_K . c
    if constexpr (FP) return fma(c.c2, at(K, 2), fma(c.c1, at(K, 1), c.c0 * at(K, 0)));
    else return dot3_64_s32(c.c0, c.c1, c.c2, at(K, 0), at(K, 1), at(K, 2));
  };
  auto neg_of = [](minor_t x) __attribute__((always_inline)) -> minor_t { if constexpr (FP) return -x; else return (u64)0 - x; };
  auto word = [](minor_t x) __attribute__((always_inline)) -> unsigned {
    if constexpr (FP) return (unsigned)__double2hiint(x) & 0x7fffffffu;
    else return (unsigned)x | ((unsigned)(x >> 32) << 1);
  };
  auto top = [](minor_t x) __attribute__((always_inline)) -> unsigned {
    if constexpr (FP) return (unsigned)__double2hiint(x); else return (unsigned)(x >> 32);
  };
  const unsigned clear_word = FP ? (unsigned)__double2hiint(clear) : 0u;
  unsigned h0 = 0, h1 = 0, g0 = 0, g1 = 0, tested = 0;
  const vec3 c15 = cross0(std::integral_constant<int, 15>{});                  // (every group's last m3)
  fan_for<10>([&](auto GC) __attribute__((always_inline)) {
    constexpr int G = decltype(GC)::value;
    constexpr int M2 = k_fan_groups.m2[G], NSUB = k_fan_groups.nsub[G], NSUP = k_fan_groups.nsup[G];
    const vec3 c2 = cross0(std::integral_constant<int, M2>{});
    minor_t n3s[NSUB];
    fan_for<NSUB>([&](auto AC) __attribute__((always_inline)) {
      constexpr int a = decltype(AC)::value;
      n3s[a] = neg_of(dot(c2, std::integral_constant<int, k_fan_groups.sub[G][a]>{}));
    });
    fan_for<NSUP>([&](auto BC) __attribute__((always_inline)) {
      constexpr int b = decltype(BC)::value;
      constexpr int M3 = k_fan_groups.sup[G][b];
      const minor_t n1 = dot(c2, std::integral_constant<int, M3>{});
      vec3 c3 = c15;
      if constexpr (M3 != 15) c3 = cross0(std::integral_constant<int, M3>{});
      fan_for<NSUB>([&](auto AC) __attribute__((always_inline)) {
        constexpr int a = decltype(AC)::value;
        constexpr int M1 = k_fan_groups.sub[G][a];
        constexpr int T = k_fan_groups.type_of[M1][M2][M3];
        static_assert(T >= 0 && T < 60, "a chain of the fan");
        constexpr unsigned tm = 1u | (1u << M1) | (1u << M2) | (1u << M3);
        constexpr bool ordinal = k_fan4.ordinal[T] != 0;
        constexpr unsigned bit = 1u << (T & 31);
        bool active = (ordinal ? do_ord : do_int) && !(inv & tm);
        if (cull) {
          const bool same = (pos[0] & tm) == tm || (neg[0] & tm) == tm || (pos[1] & tm) == tm || (neg[1] & tm) == tm || (pos[2] & tm) == tm || (neg[2] & tm) == tm;
          active = active && !same;
        }
        if (active) {
          tested ++;
          // (as in fan_of_corner3: C1 = n1, C2 = -n2, C3 = n3 first; the simplex's own determinant n0 only where they agree)
          const minor_t n3 = n3s[a];
          const minor_t n2 = neg_of(dot(c3, std::integral_constant<int, M1>{}));
          const unsigned w1 = top(n1), w2 = top(n2), w3 = top(n3);
          const bool sure3 = min(min(word(n1), word(n2)), word(n3)) > clear_word;
          const bool agree3 = (int)(~(w1 ^ w2) | (w1 ^ w3)) >= 0;
          unsigned is_hit = 0, is_unsure = sure3 ? 0u : bit;
          if (sure3 && agree3) {
            // n0 = det(X_m1, X_m2, X_m3) = X_m1 . (X_m2 x X_m3)
            const comp_t i0 = at(std::integral_constant<int, M2>{}, 0), i1 = at(std::integral_constant<int, M2>{}, 1), i2 = at(std::integral_constant<int, M2>{}, 2);
            const comp_t j0 = at(std::integral_constant<int, M3>{}, 0), j1 = at(std::integral_constant<int, M3>{}, 1), j2 = at(std::integral_constant<int, M3>{}, 2);
            const vec3 cx{minor(i1, j2, i2, j1), minor(i2, j0, i0, j2), minor(i0, j1, i1, j0)};
            const minor_t n0 = dot(cx, std::integral_constant<int, M1>{});
            const minor_t d = (n1 - n0) + (n3 - n2);
            const bool sure = min(word(n0), word(d)) > clear_word;
            const unsigned w0 = top(n0), wd = top(d);
            is_hit = (sure && (int)(~(w0 ^ wd) | (w1 ^ wd)) >= 0) ? bit : 0u;
            is_unsure = sure ? 0u : bit;
          }
          if (T < 32) { h0 |= is_hit; g0 |= is_unsure; } else { h1 |= is_hit; g1 |= is_unsure; }
        }
      });
    });
    __builtin_amdgcn_sched_barrier(0);        // (a group's values die with it: no instruction of the next group is moved up into this one)
  });
  fan_result r;
  r.hits[0] = h0; r.hits[1] = h1; r.unsure[0] = g0; r.unsure[1] = g1; r.tested = tested;
  return r;
}

// The 2D+t fan of one corner on one lane: 12 triangles over the 8 vertices of the corner's space-time cube.  A triangle is a chain
// 0 < m1 < m2; of its three "vertex replaced by the origin" determinants two contain the corner -- det(X_0, X_a), seven of them for the
// whole fan -- and one, det(X_m1, X_m2), is its own: C0 = det(X_m1, X_m2), C1 = -det(X_0, X_m2), C2 = det(X_0, X_m1), D = their sum
// (origin_in_simplex2).  Integer: components below 2^31, minors from 32 x 32 -> 64 multiplies, D in Z / 2^64 as the reference has it.  fp64:
// components below 2^25 -- products below 2^50, every value an integer below 2^53: the evaluation is EXACT, and only a true zero is "unsure".
template <class comp_t, class At>
__device__ __forceinline__ fan_result fan_of_corner2(At &&at, unsigned inv, const unsigned (&pos)[3], const unsigned (&neg)[3],
                                                     bool do_ord, bool do_int, int cull, double clear)
{
  constexpr bool FP = std::is_same<comp_t, double>::value;
  using minor_t = std::conditional_t<FP, double, u64>;
  auto minor = [&](auto I, auto J) __attribute__((always_inline)) -> minor_t {          // det(X_i, X_j)
    const comp_t a = at(I, 0), b = at(J, 1), c = at(I, 1), d = at(J, 0);
    if constexpr (FP) return fma(a, b, -(c * d));
    else return (u64)((i64)a * (i64)b) - (u64)((i64)c * (i64)d);
  };
  minor_t M[8];                                                // det(X_0, X_a)
  M[0] = minor_t(0);
  fan_for<7>([&](auto IC) __attribute__((always_inline)) {
    constexpr int a = decltype(IC)::value + 1;
    M[a] = minor(std::integral_constant<int, 0>{}, std::integral_constant<int, a>{});
  });
  auto word = [](minor_t x) __attribute__((always_inline)) -> unsigned {
    if constexpr (FP) return (unsigned)__double2hiint(x) & 0x7fffffffu;
    else return (unsigned)x | ((unsigned)(x >> 32) << 1);
  };
  auto top = [](minor_t x) __attribute__((always_inline)) -> unsigned {
    if constexpr (FP) return (unsigned)__double2hiint(x); else return (unsigned)(x >> 32);
  };
  const unsigned clear_word = FP ? (unsigned)__double2hiint(clear) : 0u;
  unsigned h0 = 0, g0 = 0, tested = 0;
  fan_for<12>([&](auto IC) __attribute__((always_inline)) {
    constexpr int T = decltype(IC)::value;
    constexpr int m1 = k_fan3.vert[T][1], m2 = k_fan3.vert[T][2];
    constexpr unsigned tm = 1u | (1u << m1) | (1u << m2);
    constexpr bool ordinal = k_fan3.ordinal[T] != 0;
    constexpr unsigned bit = 1u << T;
    bool active = (ordinal ? do_ord : do_int) && !(inv & tm);
    if (cull) {
      const bool same = (pos[0] & tm) == tm || (neg[0] & tm) == tm || (pos[1] & tm) == tm || (neg[1] & tm) == tm;
      active = active && !same;
    }
    if (active) {
      tested ++;
      // C1 = -M[m2] and C2 = M[m1] first: clear of zero and of different signs -> outside, whatever C0 and D are
      const minor_t n1 = M[m2], n2 = M[m1];
      const unsigned w1 = top(n1), w2 = top(n2);
      const bool sure2 = min(word(n1), word(n2)) > clear_word;
      const bool agree2 = (int)(w1 ^ w2) < 0;                    // sign(-n1) == sign(n2)
      unsigned is_hit = 0, is_unsure = sure2 ? 0u : bit;
      if (sure2 && agree2) {
        const minor_t n0 = minor(std::integral_constant<int, m1>{}, std::integral_constant<int, m2>{});
        const minor_t d = (n0 - n1) + n2;
        const bool sure = min(word(n0), word(d)) > clear_word;
        const unsigned w0 = top(n0), wd = top(d);
        is_hit = (sure && (int)((w0 ^ wd) | (w2 ^ wd)) >= 0) ? bit : 0u;
        is_unsure = sure ? 0u : bit;
      }
      h0 |= is_hit; g0 |= is_unsure;
    }
  });
  fan_result r;
  r.hits[0] = h0; r.hits[1] = 0; r.unsure[0] = g0; r.unsure[1] = 0; r.tested = tested;
  return r;
}

// A staged vertex, the common case first: |trunc(v * factor)| < 2^31 on every component.  Then the quantised value is one multiply, one
// v_trunc_f64 and one v_cvt_i32_f64 away, its sign bits and its magnitude class are compares on the truncated double, and the double
// itself is what the fp64 fan reads -- against (int64_t)(v * factor) with its range check, done in software on this part, then int64
// compares and an int64 -> double conversion for the same three facts (590 instructions per vertex in round 3's kernel, which a tile
// pays 3.3 times per corner).  NaN, Inf and anything outside int32 fail the first compare and take classify_value as before: same q, same
// mask byte, bit for bit (quantize(), cp_device.hpp; critical_point_tracker_3d_regular.hh:453-464).
// tq[j] = (double)q[j]; narrow / mid / small: running ANDs over the tile (|q| fits int32 / < 2^19 (3D) or 2^25 (2D) / < 2^16 or 2^25).
template <int ND>
__device__ __forceinline__ unsigned char stage_value(const double *v, double factor, i64 q[ND], double tq[ND], bool &narrow, bool &mid, bool &small)
{
  constexpr double kMid = ND == 3 ? 524288.0 : 33554432.0, kSmall = ND == 3 ? 65536.0 : 33554432.0;
  double t[ND];
  bool fast = true;
#pragma unroll
  for (int j = 0; j < ND; j ++) { t[j] = trunc(v[j] * factor); fast = fast && fabs(t[j]) < 2147483648.0; }
  if (fast) {
    unsigned char mk = 0;
    bool big = false;
#pragma unroll
    for (int j = 0; j < ND; j ++) {
      const int qi = (int)t[j];                                // (exact: t is an integer below 2^31)
      q[j] = (i64)qi;
      tq[j] = (double)qi;                                      // (+0.0 for a zero, whatever the sign of v)
      const double a = fabs(t[j]);
      if (t[j] > 0.0) mk |= (unsigned char)(1u << j);
      if (t[j] < 0.0) mk |= (unsigned char)(8u << j);
      big = big || a >= (double)safe_m<ND>();
      mid = mid && a < kMid; small = small && a < kSmall;
    }
    if (big) mk &= (unsigned char)~0x3fu;
    return mk;
  }
  const unsigned char mk = classify_value<ND>(v, factor, q);
#pragma unroll
  for (int j = 0; j < ND; j ++) {
    tq[j] = (double)q[j];
    narrow = narrow && fits_s32(q[j]);
    const u64 aq = (u64)(q[j] < 0 ? -q[j] : q[j]);
    mid = mid && aq < (1ull << (ND == 3 ? 19 : 25)); small = small && aq < (1ull << (ND == 3 ? 16 : 25));
  }
  return mk;
}

// FORM 0: (corner, type) pairs over the lanes for every tile.  FORM 1 (3D, robust test): tiles whose components fit in 32 bits take the
// integer fan, the others the pairs.  FORM 2: tiles with |component| < 2^19 take the fp64 fan, else as FORM 1.  The host picks the form
// from what it knows of the slices' magnitudes (launch_tile); every form is correct on every tile -- the forms differ in registers.
//
// Round 6: a workgroup keeps its tile for ALL the steps of the launch (TileParams::steps, nsteps -- the requests of one batch in time
// order).  Slice t + 1 of step t is slice t of step t + 1: where the next step reads the same array under the same factor, its staged
// vertices -- quantised components, doubles, mask bytes, magnitude flags -- move from slot 1 to slot 0 inside LDS (a dozen 8-byte copies
// per lane; swapping the slots' roles instead made every LDS address of the fan a run-time one and cost it its registers);
// a step stages ONE slice instead of two (round 5: both, every step: 14.5 k of a wavefront's 34.6 k cycles), and the statistics leave once
// per workgroup instead of once per step.  The block of S a slice's gradients are taken from is loaded by rows -- 13 rows of 19 values per
// round over 247 lanes, row and column fixed per lane -- instead of by a linear index that every round took apart again.
// Tried on top of this in round 6 and not kept: the NEXT step's block of S on its way while the fan runs.  As LDS-DMA (buffer_load ... lds
// behind the staging barrier, no registers): the compiler puts a vmcnt(0) in front of the next LDS read it generates, which is the fan's
// first -- 25.7 -> 30.6 ms on 256^3 x 16.  As plain loads issued behind the fan and committed at the top of the next step: 27.3 ms.  What
// a step waits for is not that round trip.
// PROBE (TileParams::repeat > 1: ftkx_debug_tile_repeat, bench.py's int-VALU yardstick): the fan phase alone is run that many times on the
// staged tile -- the rate of the predicate arithmetic without staging, lists and records.
template <int ND, int FORM, bool PROBE = false>
__global__ __launch_bounds__(kThreads, FORM == 0 ? 1 : FTKX_FAN_WAVES) void tile_kernel(const TileParams p)
{
  using cfg = tile_cfg<ND>;
  constexpr int N = ND + 1;
  constexpr int HX = cfg::TX + 1, HY = cfg::TY + 1, HZ = (ND == 3) ? cfg::TZ + 1 : 1;
  constexpr int NH = HX * HY * HZ;
  constexpr int NORD = fan_table<N>::NORD, NINT = fan_table<N>::NINT;
  static_assert(cfg::TX * cfg::TY * cfg::TZ == kThreads, "one corner per lane");

  // 3D, scalar input: the gradients of a slice's 17 x 5 x 5 vertices read a 19 x 7 x 7 block of S -- loaded once into LDS with every
  // load of a lane in flight together (one round trip), instead of six dependent global loads per vertex and 2.7 reads per value
  constexpr bool S_BLOCK = ND == 3;
  constexpr int SX = cfg::TX + 3, SY = cfg::TY + 3, SZ = cfg::TZ + 3, NS = S_BLOCK ? SX * SY * SZ : 1;
  constexpr int kItems = 512;
  __shared__ i64 s_vf[2][ND][NH];                      // quantised components, one array per (slot, component)
  __shared__ double s_vd[FORM >= 2 ? 2 * ND * NH : 1]; // FORM 2: the same as doubles (exact below 2^53; read where the tile is below 2^19)
  __shared__ double s_s[NS];                           // the block of S of the slice being staged
  __shared__ unsigned char s_mask[2][NH];
  __shared__ unsigned s_tab[fan_table<N>::NTYPES];     // the vertex masks of a type packed in one word
  __shared__ unsigned short s_list[2][kThreads];       // surviving corners: [0] ordinal sweep, [1] interval sweep
  __shared__ unsigned s_cnt[2];
  __shared__ unsigned s_wflags[2][kThreads / 64];      // per slot and wavefront: the magnitude flags of the vertices it staged
  __shared__ unsigned short s_items[FORM >= 1 ? kItems : 1];   // fan forms: (lane, type) of the simplices the fan was not sure of
  __shared__ unsigned s_nitems, s_stat[2];

  const int tid = threadIdx.x;
  const fan_table<N> &fan = dev_fan<ND>();
  const Mesh &m = p.m;
#ifdef FTKX_TILE_STAMPS
  unsigned long long stamp_ = __builtin_readcyclecounter(), phase_[6] = {0, 0, 0, 0, 0, 0};
#undef TILE_STAMP
#define TILE_STAMP(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); phase_[k] += now_ - stamp_; stamp_ = now_; } while (0)
#endif

  // workgroup -> tile.  Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share an L2): give each XCD a
  // contiguous run of tiles so that neighbouring tiles' shared halo vertices hit the same L2.
  const unsigned nblocks = gridDim.x;
  unsigned b = blockIdx.x;
  {
    const unsigned per = nblocks / 8, rem = nblocks % 8, xcd = b % 8, k = b / 8;
    b = xcd * per + (xcd < rem ? xcd : rem) + k;
  }
  const int tile[3] = {(int)(b % p.ntiles[0]), (int)((b / p.ntiles[0]) % p.ntiles[1]), (int)(b / (p.ntiles[0] * p.ntiles[1]))};
  const int origin[3] = {m.core_st[0] + tile[0] * cfg::TX, m.core_st[1] + tile[1] * cfg::TY, (ND == 3) ? m.core_st[2] + tile[2] * cfg::TZ : 0};

  if (tid < fan_table<N>::NTYPES) {
    unsigned w = 0;
    for (int i = 0; i < N; i ++) w |= (unsigned)fan.vert[tid][i] << (8 * i);
    s_tab[tid] = w;
  }
  if (tid < 2) s_stat[tid] = 0;

  const bool from_block = S_BLOCK && m.scalar_mode;
  // this lane's corner
  const int cx = tid % cfg::TX, cy = (tid / cfg::TX) % cfg::TY, cz = tid / (cfg::TX * cfg::TY);
  bool in_core = true;
  {
    const int csp[3] = {origin[0] + cx, origin[1] + cy, origin[2] + cz};
    for (int d = 0; d < ND; d ++) in_core = in_core && csp[d] < m.core_st[d] + m.core_sz[d];
  }
  const int hbase = cx + HX * (cy + HY * cz);

  // ---- one slice into one slot: mask bytes, quantised components (+ doubles), the magnitude flags of what this wavefront staged ----
  // (tl = the lane's index seen through an empty asm in every step: what the staging derives from it -- rows, vertices, LDS addresses, bounds
  // checks -- is loop-invariant, and hoisted out of the step loop it would sit in registers across the fan, which has none to spare)
  auto stage_slice = [&](const Fields &f, int sl, int slot, const int tl) {
    if constexpr (S_BLOCK) {
      if (from_block) {
        // the block by rows: lane -> (row0, col) once, 13 rows a round; row = y + SY * z of the block
        constexpr int RPR = kThreads / SX, ROUNDS = (SY * SZ + RPR - 1) / RPR;
        const int row0 = tl / SX, col = tl - row0 * SX;
        const int i = origin[0] - 1 + col - m.ext_st[0];
        const bool lane_on = row0 < RPR && i >= 0 && i < m.ext_sz[0];
        const double *S = f.S[sl];
        double got[ROUNDS];
#pragma unroll
        for (int r = 0; r < ROUNDS; r ++) {
          const int row = row0 + r * RPR, y = row % SY, z = row / SY;
          const int j = origin[1] - 1 + y - m.ext_st[1], k = origin[2] - 1 + z - m.ext_st[2];
          const bool in = lane_on && row < SY * SZ && j >= 0 && j < m.ext_sz[1] && k >= 0 && k < m.ext_sz[2];
          got[r] = in ? S[arr_index<3>(m, i, j, k)] : 0.0;
        }
#pragma unroll
        for (int r = 0; r < ROUNDS; r ++) {
          const int row = row0 + r * RPR;
          if (row0 < RPR && row < SY * SZ) s_s[col + SX * row] = got[r];
        }
        __syncthreads();
      }
    }
    TILE_STAMP(0);
    bool mine_narrow = true, mine_mid = true, mine_small = true;
    for (int hv = tl; hv < NH; hv += kThreads) {
      const int hx = hv % HX, hy = (hv / HX) % HY, hz = hv / (HX * HY);
      const int vx[3] = {origin[0] + hx, origin[1] + hy, origin[2] + hz};
      i64 q[ND];
      double tq[ND];
      unsigned char mk = kInvalid;
      for (int j = 0; j < ND; j ++) { q[j] = 0; tq[j] = 0.0; }
      if (vertex_usable<ND>(m, vx)) {
        double g[3] = {0.0, 0.0, 0.0};
        if (from_block) {
          // gradient3D of ndarray/grad.hh out of the block: the same operations as gradient_at on the same values
          const int i = vx[0] - m.ext_st[0], j = vx[1] - m.ext_st[1], k = vx[2] - m.ext_st[2];
          if (i >= 1 && i < m.ext_sz[0] - 1 && j >= 1 && j < m.ext_sz[1] - 1 && k >= 1 && k < m.ext_sz[2] - 1) {
            const double *c = &s_s[(hx + 1) + SX * ((hy + 1) + SY * (hz + 1))];
            g[0] = 0.5 * (c[1] - c[-1]);
            g[1] = 0.5 * (c[SX] - c[-SX]);
            g[2] = 0.5 * (c[SX * SY] - c[-SX * SY]);
          }
        } else vector_at<ND>(m, f.S[sl], f.V[sl], vx[0] - m.ext_st[0], vx[1] - m.ext_st[1], ND == 3 ? vx[2] - m.ext_st[2] : 0, g);
        mk = stage_value<ND>(g, f.factor, q, tq, mine_narrow, mine_mid, mine_small);
      }
      s_mask[slot][hv] = mk;
      for (int j = 0; j < ND; j ++) {
        s_vf[slot][j][hv] = q[j];
        if constexpr (FORM >= 2) s_vd[(slot * ND + j) * NH + hv] = tq[j];
      }
    }
    // all quantised components fit in 32 bits: the integer test takes its cheaper multiplies -- same values, see cp_device.hpp; below
    // 2^19, 2^16: the 3D fan decides signs in double precision, fan_of_corner3.  One word per slot and wavefront.
    const unsigned bits = (__all(mine_narrow) ? 1u : 0u) | (__all(mine_mid) ? 2u : 0u) | (__all(mine_small) ? 4u : 0u);
    if ((tl & 63) == 0) s_wflags[slot][tl >> 6] = bits;
    TILE_STAMP(1);
  };

  bool have_next = false;                              // slot 1 holds `prev_*` staged under prev_factor
  const double *prev_S = nullptr, *prev_V = nullptr;
  double prev_factor = 0.0;
  unsigned tested = 0, kept_sum = 0;

  for (int si = 0; si < p.nsteps; si ++) {
    const Fields &f = p.steps[si];                     // (in global memory, read with scalar loads: a copy would live in scratch -- the lambdas below take its address)
    const int step = p.step + si;
    const bool need_next = (f.scope_mask & FTKX_SCOPE_INTERVAL) != 0;
    // (block-uniform) the slice the last step staged as its slice 1 is this step's slice 0, under the same factor: the slots swap
    const bool reuse0 = have_next && f.S[0] == prev_S && f.V[0] == prev_V && f.factor == prev_factor;
    __syncthreads();                                   // (the step before is through with the lists, the counters and the slots; si = 0: s_tab, s_stat)
    int tl = tid;
    asm volatile("" : "+v"(tl));
    if (tl < 2) s_cnt[tl] = 0;
    if (tl == 2) s_nitems = 0;
    if (reuse0) {
      // slot 1 -> slot 0.  (Every lane moves the entries it stages itself -- the same hv below -- so its later stores to slot 1 are behind its
      // loads here in program order; the other lanes read either slot only behind the barrier after the staging.)
      for (int hv = tl; hv < NH; hv += kThreads) {
        s_mask[0][hv] = s_mask[1][hv];
        for (int j = 0; j < ND; j ++) {
          s_vf[0][j][hv] = s_vf[1][j][hv];
          if constexpr (FORM >= 2) s_vd[j * NH + hv] = s_vd[(ND + j) * NH + hv];
        }
      }
      if ((tl & 63) == 0) s_wflags[0][tl >> 6] = s_wflags[1][tl >> 6];
    } else stage_slice(f, 0, 0, tl);
    if (need_next) {
      if (!reuse0 && from_block) __syncthreads();      // (slice 0's vertices have been taken from s_s)
      stage_slice(f, 1, 1, tl);
    }
    have_next = need_next; prev_S = f.S[1]; prev_V = f.V[1]; prev_factor = f.factor;
    __syncthreads();
    unsigned tile_bits = 7u;
    for (int w = 0; w < kThreads / 64; w ++) { tile_bits &= s_wflags[0][w]; if (need_next) tile_bits &= s_wflags[1][w]; }
    const bool narrow = (tile_bits & 1u) != 0;
    TILE_STAMP(2);
    const bool fan_int = FORM >= 1 && narrow && (ND == 2 || m.robust) && p.fan >= 1;
    const bool fan_fp = FORM >= 2 && fan_int && p.fan >= 2 && (tile_bits & 2u);
    const bool small = fan_fp && (tile_bits & 4u);
    const unsigned char *mask0 = s_mask[0], *mask1 = s_mask[1];

    // ---- cull: one corner per lane ----
    bool keep_o = false, keep_i = false;
    {
      unsigned and0 = 0x3f, and1 = 0x3f;
      for (int c = 0; c < (1 << ND); c ++) {
        const int off = (c & 1) + HX * (((c >> 1) & 1) + HY * ((c >> 2) & 1));
        // vertices no simplex may use (outside the domain, non-finite) are neutral for the sign argument
        const unsigned m0 = mask0[hbase + off];
        and0 &= (m0 & (kInvalid | kNonFinite)) ? 0x3fu : m0;
        if (need_next) { const unsigned m1 = mask1[hbase + off]; and1 &= (m1 & (kInvalid | kNonFinite)) ? 0x3fu : m1; }
      }
      keep_o = in_core && (f.scope_mask & FTKX_SCOPE_ORDINAL) && !(p.cull && (and0 & 0x3f));
      keep_i = in_core && need_next && !(p.cull && (and0 & and1 & 0x3f));
    }
    const unsigned long long ballot_o = __ballot(keep_o), ballot_i = __ballot(keep_i);
    if ((tid & 63) == 0) kept_sum += (unsigned)__popcll(need_next ? ballot_i : ballot_o);
    if constexpr (FORM >= 1) {
      // ---- test, one corner per lane (fan_of_corner3 / fan_of_corner2) ----
      if (fan_int && p.fan != 9) {
        u64 hits = 0, unsure = 0;
        constexpr int NV = 1 << N;                                  // vertices of the corner's space-time hypercube
        int corner[N];
        corner[0] = origin[0] + cx; corner[1] = origin[1] + cy;
        if (ND == 3) corner[2] = origin[2] + cz;
        corner[ND] = f.t;
        // (vertex v of the corner's hypercube: bit d = one step along axis d, bit ND = the next slice)
        auto offset = [](int v) constexpr { return (v & 1) + HX * (((v >> 1) & 1) + (ND == 3 ? HY * ((v >> 2) & 1) : 0)); };
        if (ballot_o | ballot_i) {                                 // (a wavefront nothing of which survived its cull: nothing to do)
          unsigned inv = 0, pos[3] = {0, 0, 0}, neg[3] = {0, 0, 0};
#pragma unroll
          for (int v = 0; v < NV; v ++) {
            unsigned mk = kInvalid;
            if ((v >> ND) == 0 || need_next) mk = ((v >> ND) ? mask1 : mask0)[hbase + offset(v)];
            if (mk & (kInvalid | kNonFinite)) inv |= 1u << v;
#pragma unroll
            for (int c = 0; c < ND; c ++) { if (mk & (1u << c)) pos[c] |= 1u << v; if (mk & (8u << c)) neg[c] |= 1u << v; }
          }
          // (component c of vertex v; slice 1 of an ordinal-only request is not staged: whatever is read there only enters simplices that
          // `inv` switches off)
          auto run_fan = [&](const int hb_) __attribute__((always_inline)) -> fan_result {
            if (FORM >= 2 && fan_fp) {
              const double *base = s_vd + hb_;
              auto at = [&](auto V, int c) __attribute__((always_inline)) { constexpr int v = decltype(V)::value; return base[((v >> ND) * ND + c) * NH + offset(v)]; };
              if constexpr (ND == 3) return fan_of_corner3<double>(at, inv, pos, neg, keep_o, keep_i, p.cull, small ? 0.5 : 4096.0);
              else return fan_of_corner2<double>(at, inv, pos, neg, keep_o, keep_i, p.cull, 0.5);
            } else {
              const i64 *base = &s_vf[0][0][0] + hb_;
              auto at = [&](auto V, int c) __attribute__((always_inline)) { constexpr int v = decltype(V)::value; return (int)base[((v >> ND) * ND + c) * NH + offset(v)]; };
              if constexpr (ND == 3) return fan_of_corner3<int>(at, inv, pos, neg, keep_o, keep_i, p.cull, 0.0);
              else return fan_of_corner2<int>(at, inv, pos, neg, keep_o, keep_i, p.cull, 0.0);
            }
          };
          fan_result fr;
          if constexpr (PROBE) {
            for (int rep_ = 0; rep_ < p.repeat; rep_ ++) { int hb_ = hbase; asm volatile("" : "+v"(hb_)); fr = run_fan(hb_); }
          } else fr = run_fan(hbase);
          tested += fr.tested;
          hits = (u64)fr.hits[0] | ((u64)fr.hits[1] << 32); unsure = (u64)fr.unsure[0] | ((u64)fr.unsure[1] << 32);
        }
        TILE_STAMP(3);
        // A value that is zero / INT64_MIN (fp64: not clear of zero): the integer test and the literal cascade on the vertices as staged.
        // One simplex in thousands, but a long computation: the (lane, type) pairs of the whole tile go on a list and are dealt to the lanes
        // again (a lane that walked its own would hold its wavefront for each of them).
        auto corner_of = [&](int lane_tid, int (&lc)[N]) -> int {
          const int lx = lane_tid % cfg::TX, ly = (lane_tid / cfg::TX) % cfg::TY, lz = lane_tid / (cfg::TX * cfg::TY);
          lc[0] = origin[0] + lx; lc[1] = origin[1] + ly;
          if (ND == 3) lc[2] = origin[2] + lz;
          lc[ND] = f.t;
          return lx + HX * (ly + HY * lz);
        };
        auto resolve = [&](int lane_tid, int type) -> bool {
          int lc[N];
          const int hb = corner_of(lane_tid, lc);
          const unsigned tab = s_tab[type];
          u64 X[N][ND]; int ids[N];
          for (int i = 0; i < N; i ++) {
            const unsigned vm = (tab >> (8 * i)) & 0xffu;
            const int hidx = hb + (vm & 1) + HX * (((vm >> 1) & 1) + (ND == 3 ? HY * ((vm >> 2) & 1) : 0));
            for (int j = 0; j < ND; j ++) X[i][j] = (u64)s_vf[(vm >> ND) & 1][j][hidx];
            ids[i] = vertex_id<ND>(m, lc, vm);
          }
          int r;
          if constexpr (ND == 2) r = origin_in_simplex2_try(X, true); else r = origin_in_simplex3_try(X, true);
          return r < 0 ? sos_origin_in_simplex_resolved<ND>(X, ids) : r != 0;
        };
        for (;;) {                                                 // (one round unless the tile has more than kItems of them)
          while (__any(unsure != 0)) {                             // append (wave-uniform trip count)
            const bool have = unsure != 0;
            const int type = have ? __ffsll((long long)unsure) - 1 : 0;
            const unsigned long long hb = __ballot(have);
            const int leader = __ffsll((long long)hb) - 1;
            unsigned base = 0;
            if ((tid & 63) == leader) base = atomicAdd(&s_nitems, (unsigned)__popcll(hb));
            base = __shfl(base, leader);
            const unsigned slot = base + (unsigned)__popcll(hb & ((1ull << (tid & 63)) - 1ull));
            if (have && slot < (unsigned)kItems) { s_items[slot] = (unsigned short)((tid << 6) | type); unsure &= unsure - 1; }
            if (base + (unsigned)__popcll(hb) > (unsigned)kItems) break;   // (the list is full -- the count says so to everybody: what is left waits for the next round)
          }
          __syncthreads();
          const unsigned appended = s_nitems, nitems = appended < (unsigned)kItems ? appended : (unsigned)kItems;
          for (unsigned base = 0; base < nitems; base += kThreads) {
            const unsigned it = base + tid;
            bool hit = false;
            u64 desc = 0;
            if (it < nitems) {
              const unsigned item = s_items[it];
              const int lane_tid = (int)(item >> 6), type = (int)(item & 63u);
              hit = resolve(lane_tid, type);
              int lc[N];
              (void)corner_of(lane_tid, lc);
              desc = core_linear<ND>(m, lc) | ((u64)type << kPassTypeShift) | ((u64)step << kPassStepShift);
            }
            emit_pass(m, hit, desc);
          }
          if (appended <= (unsigned)kItems) break;                 // (block-uniform: everybody read the same count)
          __syncthreads();
          if (tid == 0) s_nitems = 0;
          __syncthreads();
        }
        // the simplices that passed, appended for record_kernel: ONE reservation per wavefront for all of its lanes' hits (a lane has up to
        // 60 of them in the overflow regime, where the reference's wrapped determinants "hit" everywhere; a reservation per hit type -- the
        // loop this replaces -- was up to 60 returning atomics one behind the other per wavefront: 113 M records, 66 ms per 256^3 x 4)
        const u64 lin = in_core ? core_linear<ND>(m, corner) : 0ull;
        if (__any(hits != 0)) {
          const unsigned mine = (unsigned)__popcll(hits);
          unsigned incl = mine;
          for (int o = 1; o < 64; o <<= 1) { const unsigned v = __shfl_up(incl, o); if ((tid & 63) >= o) incl += v; }
          const unsigned total = __shfl(incl, 63);
          u64 slot0 = 0;
          if ((tid & 63) == 0) slot0 = atomicAdd(&m.counters[CNT_PASS], (u64)total);
          slot0 = __shfl(slot0, 0) + (u64)(incl - mine);
          while (hits != 0) {
            const int type = __ffsll((long long)hits) - 1;
            hits &= hits - 1;
            if (slot0 < m.capacity) m.pass[slot0] = lin | ((u64)type << kPassTypeShift) | ((u64)step << kPassStepShift);
            slot0 ++;
          }
        }
      }
    }
    // the surviving corners as lists, for the (corner, type) pairs below
    const bool pairs = !(fan_int || p.fan == 9);                 // (block-uniform)
    if (pairs) {
      const int lane = tid & 63;
      const unsigned long long below = (1ull << lane) - 1ull;
      unsigned base_o = 0, base_i = 0;
      if (lane == 0) {
        if (ballot_o) base_o = atomicAdd(&s_cnt[0], (unsigned)__popcll(ballot_o));
        if (ballot_i) base_i = atomicAdd(&s_cnt[1], (unsigned)__popcll(ballot_i));
      }
      base_o = __shfl(base_o, 0);
      base_i = __shfl(base_i, 0);
      if (keep_o) s_list[0][base_o + __popcll(ballot_o & below)] = (unsigned short)tid;
      if (keep_i) s_list[1][base_i + __popcll(ballot_i & below)] = (unsigned short)tid;
      __syncthreads();
    }
    const unsigned n_o = pairs ? s_cnt[0] : 0u, n_i = pairs ? s_cnt[1] : 0u;
    // ---- test: (corner, type) pairs over all lanes ----
    const unsigned items_o = n_o * NORD, total = items_o + n_i * NINT;
    for (unsigned base = 0; base < total; base += kThreads) {   // wave-uniform trip count: the ballot in emit_hits stays convergent
      const unsigned w = base + tid;
      bool hit = false;
      u64 desc = 0;
      if (w < total) {
        const bool ordinal = w < items_o;
        const unsigned wl = ordinal ? w : w - items_o;
        const unsigned ci = ordinal ? wl / NORD : wl / NINT;
        const unsigned it = ordinal ? wl % NORD : wl % NINT;
        const int type = ordinal ? fan.ord_types[it] : fan.int_types[it];
        const int ct = s_list[ordinal ? 0 : 1][ci];
        const int ccx = ct % cfg::TX, ccy = (ct / cfg::TX) % cfg::TY, ccz = ct / (cfg::TX * cfg::TY);
        const int hb = ccx + HX * (ccy + HY * ccz);
        const unsigned tab = s_tab[type];
        unsigned char flags[N];
        u64 X[N][ND];
        for (int i = 0; i < N; i ++) {
          const unsigned vm = (tab >> (8 * i)) & 0xffu;
          const int hidx = hb + (vm & 1) + HX * (((vm >> 1) & 1) + ((ND == 3) ? HY * ((vm >> 2) & 1) : 0));
          const int hsl = (vm >> ND) & 1;
          flags[i] = s_mask[hsl][hidx];
          for (int j = 0; j < ND; j ++) X[i][j] = (u64)s_vf[hsl][j][hidx];
        }
        int corner[N];
        corner[0] = origin[0] + ccx; corner[1] = origin[1] + ccy;
        if (ND == 3) corner[2] = origin[2] + ccz;
        corner[ND] = f.t;
        int ids[N]; double mu[N]; bool presolved;
        hit = simplex_inside<ND>(m, f, p.cull, corner, tab, flags, X, tested, ids, mu, &presolved, narrow);
        desc = core_linear<ND>(m, corner) | ((u64)type << kPassTypeShift) | ((u64)step << kPassStepShift);
      }
      emit_pass(m, hit, desc);
    }
    TILE_STAMP(4);
  }
  {
    // statistics, once per workgroup: one atomic per counter (same-address atomics serialise chip-wide)
    unsigned t_sum = tested;
    for (int o = 32; o > 0; o >>= 1) t_sum += __shfl_down(t_sum, o);
    if ((tid & 63) == 0) {
      if (t_sum) atomicAdd(&s_stat[0], t_sum);
      if (kept_sum) atomicAdd(&s_stat[1], kept_sum);
    }
    __syncthreads();
    u64 *slot = p.stats + 2u * (blockIdx.x & 255u);
    if (tid == 0 && s_stat[0]) atomicAdd(&slot[0], (u64)s_stat[0]);
    if (tid == 64 && s_stat[1]) atomicAdd(&slot[1], (u64)s_stat[1]);
  }
  TILE_STAMP(5);
#ifdef FTKX_TILE_STAMPS
  if ((tid & 63) == 0) {
    unsigned long long *g = g_tile_stamps + (blockIdx.x % 512u) * 8;
    for (int k = 0; k < 6; k ++) atomicAdd(&g[k], phase_[k]);
    atomicAdd(&g[7], (unsigned long long)p.nsteps);
  }
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------------------
// the tile kernels' statistics, 256 slots -> the two counters (and the slots cleared for the next batch)
__global__ __launch_bounds__(256) void tile_stats_fold_kernel(u64 *__restrict__ slots, u64 *__restrict__ counters)
{
  u64 a = slots[2 * threadIdx.x], b = slots[2 * threadIdx.x + 1];
  slots[2 * threadIdx.x] = 0; slots[2 * threadIdx.x + 1] = 0;
  for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o); b += __shfl_down(b, o); }
  if ((threadIdx.x & 63) == 0) {
    if (a) atomicAdd(&counters[CNT_SIMPLICES_TESTED], a);
    if (b) atomicAdd(&counters[CNT_CELLS_SURVIVED], b);
  }
}
void launch_tile_stats_fold(u64 *slots, u64 *counters, hipStream_t stream) { hipLaunchKernelGGL(tile_stats_fold_kernel, dim3(1), dim3(256), 0, stream, slots, counters); }

#ifdef FTKX_TILE_STAMPS
extern "C" void ftkx_debug_tile_stamps(unsigned long long *out, int reset)
{
  static unsigned long long h[512 * 8];
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_tile_stamps), sizeof(h));
  for (int k = 0; k < 8; k ++) { out[k] = 0; for (int b = 0; b < 512; b ++) out[k] += h[b * 8 + k]; }
  if (reset) { for (auto &x : h) x = 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tile_stamps), h, sizeof(h)); }
}
#endif
void launch_tile(const TileParams &p, hipStream_t stream)
{
  const unsigned nblocks = (unsigned)p.ntiles[0] * p.ntiles[1] * p.ntiles[2];
  if (nblocks == 0) return;
  if (p.repeat > 1 && p.form >= 1) {        // (ftkx_debug_tile_repeat: the fan forms with their fan phase in a loop)
    if (p.m.nd == 2) { if (p.form >= 2) hipLaunchKernelGGL((tile_kernel<2, 2, true>), dim3(nblocks), dim3(kThreads), 0, stream, p); else hipLaunchKernelGGL((tile_kernel<2, 1, true>), dim3(nblocks), dim3(kThreads), 0, stream, p); }
    else { if (p.form >= 2) hipLaunchKernelGGL((tile_kernel<3, 2, true>), dim3(nblocks), dim3(kThreads), 0, stream, p); else hipLaunchKernelGGL((tile_kernel<3, 1, true>), dim3(nblocks), dim3(kThreads), 0, stream, p); }
    return;
  }
  if (p.m.nd == 2) {
    if (p.form >= 2) hipLaunchKernelGGL((tile_kernel<2, 2>), dim3(nblocks), dim3(kThreads), 0, stream, p);
    else if (p.form == 1) hipLaunchKernelGGL((tile_kernel<2, 1>), dim3(nblocks), dim3(kThreads), 0, stream, p);
    else hipLaunchKernelGGL((tile_kernel<2, 0>), dim3(nblocks), dim3(kThreads), 0, stream, p);
  }
  else if (p.form >= 2) hipLaunchKernelGGL((tile_kernel<3, 2>), dim3(nblocks), dim3(kThreads), 0, stream, p);
  else if (p.form == 1) hipLaunchKernelGGL((tile_kernel<3, 1>), dim3(nblocks), dim3(kThreads), 0, stream, p);
  else hipLaunchKernelGGL((tile_kernel<3, 0>), dim3(nblocks), dim3(kThreads), 0, stream, p);
}

void tile_dims(int nd, int tile[3])
{
  if (nd == 2) { tile[0] = tile_cfg<2>::TX; tile[1] = tile_cfg<2>::TY; tile[2] = 1; }
  else { tile[0] = tile_cfg<3>::TX; tile[1] = tile_cfg<3>::TY; tile[2] = tile_cfg<3>::TZ; }
}

}  // namespace ftkx
