// The space-time simplex fan of a regular lattice, generated at compile time.
//
// What it must equal: the tables ftk::simplicial_regular_mesh builds at run time
// (include/ftk/mesh/simplicial_regular_mesh.hh:891-927 initialize_subdivision -> subdivide_unit_cube 620-653,
// reduce_unit_simplex 655-683, enumerate_unit_simplices 685-715, derive_ordinal_and_interval_simplices 799-831).
// Closed form used here: a reduced k-simplex of the Kuhn subdivision of the unit n-cube is a chain
//   0 = m_0 < m_1 < ... < m_k   of nested axis sets (each vertex = indicator vector of its set);
// types are numbered in lexicographic order of the vertex list, a vertex being compared as a 0/1 vector with
// axis 0 (x) MOST significant; a type is ordinal iff no vertex has the time axis (the last one).
// tests/test_host_numerics.py checks these tables against the reference's own dump.
#pragma once

namespace ftkx {

template <int N>   // N = mesh dimension: 3 for 2D+t, 4 for 3D+t; simplices of dimension N-1
struct fan_table {
  static constexpr int NV = N;                                   // vertices per (N-1)-simplex
  static constexpr int NTYPES = (N == 3) ? 12 : 60;
  static constexpr int NORD = (N == 3) ? 2 : 6;
  static constexpr int NINT = NTYPES - NORD;
  unsigned char vert[NTYPES][NV];   // vertex offset as an axis bitmask: bit a = offset along axis a (x = bit 0, time = bit N-1)
  unsigned char ordinal[NTYPES];
  unsigned char ord_types[NORD];    // scope-local index -> type id (unit_ordinal_simplex_types)
  unsigned char int_types[NINT];    // unit_interval_simplex_types
  unsigned char local_index[NTYPES];// inverse of the two lists: rank of a type inside its own scope (from_work_index, :480-493)
};

namespace detail {
// key: integer whose MSB-first bit string is the 0/1 vector (x first) -> integer order == lexicographic order
template <int N> constexpr unsigned key_to_mask(unsigned key)
{
  unsigned m = 0;
  for (int a = 0; a < N; a ++)
    if ((key >> (N - 1 - a)) & 1u) m |= 1u << a;
  return m;
}
}  // namespace detail

template <int N>
constexpr fan_table<N> make_fan()
{
  fan_table<N> f{};
  int nt = 0;
  constexpr unsigned full = 1u << N;
  // chains 0 < k1 < k2 (< k3) in increasing key order: nested loops give lexicographic order of the vertex list
  for (unsigned k1 = 1; k1 < full; k1 ++)
    for (unsigned k2 = 1; k2 < full; k2 ++) {
      if (k2 == k1 || (k2 & k1) != k1) continue;
      if (N == 3) {
        f.vert[nt][0] = 0;
        f.vert[nt][1] = (unsigned char)detail::key_to_mask<N>(k1);
        f.vert[nt][2] = (unsigned char)detail::key_to_mask<N>(k2);
        nt ++;
      } else {
        for (unsigned k3 = 1; k3 < full; k3 ++) {
          if (k3 == k2 || (k3 & k2) != k2) continue;
          f.vert[nt][0] = 0;
          f.vert[nt][1] = (unsigned char)detail::key_to_mask<N>(k1);
          f.vert[nt][2] = (unsigned char)detail::key_to_mask<N>(k2);
          f.vert[nt][N - 1] = (unsigned char)detail::key_to_mask<N>(k3);
          nt ++;
        }
      }
    }
  int no = 0, ni = 0;
  for (int t = 0; t < nt; t ++) {
    bool has_time = false;
    for (int i = 0; i < N; i ++)
      if (f.vert[t][i] & (1u << (N - 1))) has_time = true;
    f.ordinal[t] = has_time ? 0 : 1;
    if (has_time) { f.local_index[t] = (unsigned char)ni; f.int_types[ni ++] = (unsigned char)t; }
    else { f.local_index[t] = (unsigned char)no; f.ord_types[no ++] = (unsigned char)t; }
  }
  return f;
}

inline constexpr fan_table<3> k_fan3 = make_fan<3>();
inline constexpr fan_table<4> k_fan4 = make_fan<4>();

// spot checks against the documented tables
static_assert(k_fan3.ord_types[0] == 4 && k_fan3.ord_types[1] == 8, "2D+t ordinal types must be {4, 8}");
static_assert(k_fan4.ord_types[0] == 16 && k_fan4.ord_types[1] == 20 && k_fan4.ord_types[2] == 30 &&
              k_fan4.ord_types[3] == 34 && k_fan4.ord_types[4] == 46 && k_fan4.ord_types[5] == 50,
              "3D+t ordinal types must be {16, 20, 30, 34, 46, 50}");
// type 4 of the 2D+t fan is 000 010 110 (xyt): masks y, x|y
static_assert(k_fan3.vert[4][1] == 0b010 && k_fan3.vert[4][2] == 0b011, "2D+t type 4");
// type 16 of the 3D+t fan is 0000 0010 0110 1110 (xyzt): masks z, y|z, x|y|z
static_assert(k_fan4.vert[16][1] == 0b0100 && k_fan4.vert[16][2] == 0b0110 && k_fan4.vert[16][3] == 0b0111, "3D+t type 16");

// The 3 x 3 minors the 60 simplices of a 3D+t corner share.  A simplex is a chain 0 < m1 < m2 < m3 of vertex masks; three of its four
// "vertex replaced by the origin" determinants contain the corner -- det(X_0, X_a, X_b) for a pair a < b of its masks -- and there are
// only 50 such pairs in the whole fan (each used by 3.6 simplices on average); the fourth, det(X_m1, X_m2, X_m3), is the simplex's own.
// The fan grouped by the MIDDLE vertex of the chain 0 < m1 < m2 < m3 (tile_kernels.hip, fan_of_corner3_grouped): with c(m) = X_0 x X_m,
//   n1 = det(X_0, X_m2, X_m3) =  X_m3 . c(m2),   n3 = det(X_0, X_m1, X_m2) = -X_m1 . c(m2),   n2 = det(X_0, X_m1, X_m3) = -X_m1 . c(m3)
// -- a group (one m2: 10 of them, |m2| = 2 or 3) needs c(m2), at most three n1, at most six n3 and the c(m3) of its at most three m3,
// eighteen doubles live instead of the fifty shared determinants of the flat form.
struct fan_groups {
  unsigned char m2[10];
  unsigned char nsub[10], sub[10][6];   // proper non-empty subsets of m2
  unsigned char nsup[10], sup[10][3];   // proper supersets of m2
  signed char type_of[16][16][16];      // (m1, m2, m3) -> type id, -1: not a chain
};
constexpr fan_groups make_fan_groups()
{
  fan_groups g{};
  for (int a = 0; a < 16; a ++) for (int b = 0; b < 16; b ++) for (int c = 0; c < 16; c ++) g.type_of[a][b][c] = -1;
  for (int t = 0; t < 60; t ++) g.type_of[k_fan4.vert[t][1]][k_fan4.vert[t][2]][k_fan4.vert[t][3]] = (signed char)t;
  int n = 0;
  for (unsigned m = 1; m < 15; m ++) {
    int bits = 0;
    for (int a = 0; a < 4; a ++) bits += (m >> a) & 1u;
    if (bits != 2 && bits != 3) continue;
    g.m2[n] = (unsigned char)m;
    for (unsigned x = 1; x < 16; x ++) {
      if (x != m && (x & m) == x) g.sub[n][g.nsub[n] ++] = (unsigned char)x;
      if (x != m && (x & m) == m) g.sup[n][g.nsup[n] ++] = (unsigned char)x;
    }
    n ++;
  }
  return g;
}
inline constexpr fan_groups k_fan_groups = make_fan_groups();
constexpr int fan_groups_types()
{
  int n = 0;
  for (int i = 0; i < 10; i ++) for (int a = 0; a < k_fan_groups.nsub[i]; a ++) for (int b = 0; b < k_fan_groups.nsup[i]; b ++)
    if (k_fan_groups.type_of[k_fan_groups.sub[i][a]][k_fan_groups.m2[i]][k_fan_groups.sup[i][b]] >= 0) n ++;
  return n;
}
static_assert(fan_groups_types() == 60, "the ten groups cover the sixty types exactly once");

template <int N> struct fan_of;
template <> struct fan_of<3> { static constexpr const fan_table<3> &get() { return k_fan3; } };
template <> struct fan_of<4> { static constexpr const fan_table<4> &get() { return k_fan4; } };

}  // namespace ftkx
