// Pass 2 on the hit set: from the discrete critical-point records of the sweep to traced curves.
// Host-side, like the reference's own pass 2 -- it touches only the hits (<= ~0.5 % of the simplices).
//
// Reference semantics reproduced (include/ftk/...):
//   neighbourhood      two d-simplices are neighbours iff they share a (d+1)-cell: f.side_of(m) -> c.sides(m)
//                      (filters/critical_point_tracker_2d_regular.hh:189-197; mesh/simplicial_regular_mesh.hh:571-601, 717-797)
//   components         union-find over the hits (filters/critical_point_tracker.hh:688-703, basic/duf.hh)
//   curves             geometry/cc2curves.hh:10-111: nodes with more than two neighbours in the hit set are "special" and
//                      dropped; every connected component of the remaining ("ordinary") nodes is one curve, ordered by walking
//                      from its smallest element, first towards its smallest neighbour, then the other way
//   loop flag          geometry/cc2curves.hh:113-122
// "smallest" is the order of simplicial_regular_mesh_element::operator< (mesh/simplicial_regular_mesh.hh:327-337): corners compared
// as vectors with x FIRST, then the type.  The traversal below follows the same order so that each curve's point sequence equals
// the reference's (tests/test_trace.py compares against curves dumped from the real reference).
#include <algorithm>
#include "host_sort.hpp"
#include <array>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <unistd.h>
#include <vector>

#include "../../include/ftkx.h"
#include "fan_tables.hpp"

namespace {

typedef unsigned long long u64;
#define ND_OF(N) ((N) - 1)

// the tags of a record set: inside records (72 bytes apart) or as an array of their own (a caller that holds its points in another form --
// the C++ tracker -- hands over 8 bytes per record instead of building records around them)
struct TagView {
  const unsigned char *p; size_t stride;
  TagView(const ftkx_cp_t *recs) : p(reinterpret_cast<const unsigned char *>(recs) + offsetof(ftkx_cp_t, tag)), stride(sizeof(ftkx_cp_t)) {}
  TagView(const unsigned long long *tags) : p(reinterpret_cast<const unsigned char *>(tags)), stride(sizeof(unsigned long long)) {}
  u64 operator[](size_t i) const { u64 v; memcpy(&v, p + i * stride, sizeof(v)); return v; }
};

struct Elem {
  int c[4];   // corner x, y, (z,) t ; unused axes 0
  int type;
};

inline bool elem_less(const Elem &a, const Elem &b, int n)
{
  for (int d = 0; d < n; d ++) if (a.c[d] != b.c[d]) return a.c[d] < b.c[d];
  return a.type < b.type;
}

// adjacency of the simplex fan, generated from the same chain rule as fan_tables.hpp
template <int N>
struct Adjacency {
  static constexpr int NF = ftkx::fan_table<N>::NTYPES;          // (N-1)-simplex types
  static constexpr int NC = (N == 3) ? 6 : 24;                   // N-cell types (N! Kuhn simplices)
  struct Ref { int type; int off[N]; };
  std::array<std::array<Ref, N + 1>, NC> sides;                  // faces of a cell
  std::array<std::vector<Ref>, NF> side_of;                      // cells containing a face (always 2)

  Adjacency()
  {
    const ftkx::fan_table<N> &fan = ftkx::fan_of<N>::get();
    // cells: chains 0 < k1 < ... < kN = full, in lexicographic order of the vertex list (x most significant):
    // enumerate keys (MSB = x) in increasing order at every level
    std::vector<std::array<unsigned, N + 1>> cells;   // vertex axis masks (bit a = axis a)
    std::array<unsigned, N + 1> chain{};
    enumerate(1, 0u, chain, cells);
    for (int ct = 0; ct < NC; ct ++) {
      for (int drop = 0; drop <= N; drop ++) {
        unsigned v[N]; int m = 0;
        for (int i = 0; i <= N; i ++) if (i != drop) v[m ++] = cells[ct][i];
        unsigned common = ~0u;
        for (int i = 0; i < N; i ++) common &= v[i];             // axes set in every remaining vertex -> corner offset
        for (int i = 0; i < N; i ++) v[i] &= ~common;
        int ft = -1;
        for (int t = 0; t < NF && ft < 0; t ++) {
          bool same = true;
          for (int i = 0; i < N; i ++) same = same && fan.vert[t][i] == v[i];
          if (same) ft = t;
        }
        Ref r; r.type = ft;
        for (int a = 0; a < N; a ++) r.off[a] = (common >> a) & 1;
        sides[ct][drop] = r;
        Ref back; back.type = ct;
        for (int a = 0; a < N; a ++) back.off[a] = -r.off[a];
        side_of[ft].push_back(back);
      }
    }
    build_candidates();
  }

  // The records that can share a cell with a face of type t, as STATIC offsets from its corner: for each of the two cells around it
  // the cell's other faces.  All candidates share the base corner, so the element order of the reference's std::set (corner as a
  // vector, x first, then the type) is the order of (off[0], ..., off[N-1], type) -- sorted once here instead of once per record.
  struct Cand { int type; int off[N]; };
  std::array<std::vector<Cand>, NF> cand;
  void build_candidates()
  {
    for (int t = 0; t < NF; t ++) {
      std::vector<Cand> v;
      for (const auto &cell : side_of[t])
        for (const auto &face : sides[cell.type]) {
          Cand c; c.type = face.type;
          bool self = face.type == t;
          for (int a = 0; a < N; a ++) { c.off[a] = cell.off[a] + face.off[a]; self = self && c.off[a] == 0; }
          if (!self) v.push_back(c);
        }
      auto less = [](const Cand &a, const Cand &b) { for (int d = 0; d < N; d ++) if (a.off[d] != b.off[d]) return a.off[d] < b.off[d]; return a.type < b.type; };
      std::sort(v.begin(), v.end(), less);
      v.erase(std::unique(v.begin(), v.end(), [&](const Cand &a, const Cand &b) { return !less(a, b) && !less(b, a); }), v.end());
      cand[t] = v;
    }
  }

private:
  static unsigned key_to_mask(unsigned key) { unsigned m = 0; for (int a = 0; a < N; a ++) if ((key >> (N - 1 - a)) & 1u) m |= 1u << a; return m; }
  void enumerate(int depth, unsigned prev_key, std::array<unsigned, N + 1> &chain, std::vector<std::array<unsigned, N + 1>> &out)
  {
    if (depth == N + 1) { out.push_back(chain); return; }
    for (unsigned key = 0; key < (1u << N); key ++)
      if (key != prev_key && (key & prev_key) == prev_key) {
        // a maximal chain adds exactly one axis per level
        if (__builtin_popcount(key) != depth) continue;
        chain[depth] = key_to_mask(key);
        enumerate(depth + 1, key, chain, out);
      }
  }
};

template <int N>
struct Tracer {
  static constexpr int ND = N - 1;
  static constexpr int NTYPES = ftkx::fan_table<N>::NTYPES;
  const Adjacency<N> &adj;
  long long lb[3], sz[3];
  u64 prod[4];
  std::vector<std::pair<u64, int>> index;   // (tag, record) sorted by tag

  Tracer(const Adjacency<N> &a, const long long *dst, const long long *dsz) : adj(a)
  {
    prod[0] = 1;
    for (int d = 0; d < ND; d ++) { lb[d] = dst[d]; sz[d] = dsz[d]; prod[d + 1] = prod[d] * (u64)dsz[d]; }
  }

  // e.to_integer(m) in 64-bit arithmetic (mesh/simplicial_regular_mesh.hh:496-502); false if the corner is outside the mesh box
  bool encode(const Elem &e, u64 *tag) const
  {
    u64 ci = 0;
    for (int d = 0; d < ND; d ++) {
      const long long rel = e.c[d] - lb[d];
      if (rel < 0 || rel >= sz[d]) return false;
      ci += (u64)rel * prod[d];
    }
    if (e.c[ND] < 0) return false;
    ci += (u64)e.c[ND] * prod[ND];
    *tag = ci * (u64)NTYPES + (u64)e.type;
    return true;
  }
  Elem decode(u64 tag) const
  {
    Elem e; e.c[0] = e.c[1] = e.c[2] = e.c[3] = 0;
    e.type = (int)(tag % (u64)NTYPES);
    u64 ci = tag / (u64)NTYPES;
    for (int d = 0; d < ND; d ++) { e.c[d] = (int)(lb[d] + (long long)(ci % (u64)sz[d])); ci /= (u64)sz[d]; }
    e.c[ND] = (int)ci;
    return e;
  }
  // tag -> record: open addressing over a power-of-two table at most half full (a lookup is one or two probes; the binary search
  // over the sorted tags it replaces cost ~16 mispredicted branches per lookup, 6-8 lookups per record)
  struct Slot { u64 tag; long long rec; };       // one cache line access per probe (tag and record side by side)
  std::vector<Slot> h_slot;
  u64 h_mask = 0;
  static u64 mix(u64 x) { x *= 0x9e3779b97f4a7c15ull; return x ^ (x >> 29); }
  // parallel = the records are claimed slot by slot with a compare-and-swap on `rec` (tags are unique and nobody looks anything up
  // before the table is complete, so a slot whose tag is still being written only has to read as taken)
  void build_hash(const ftkx_cp_t *recs, size_t n, bool parallel = false);
  template <class Tags> void build_hash_tags(const Tags &tags, size_t n, bool parallel);
  int find_tag(u64 tag) const
  {
    for (u64 p = mix(tag) & h_mask; h_slot[p].rec >= 0; p = (p + 1) & h_mask) if (h_slot[p].tag == tag) return (int)h_slot[p].rec;
    return -1;
  }
  int find(const Elem &e) const
  {
    u64 tag;
    if (!encode(e, &tag)) return -1;
    return find_tag(tag);
  }
  // the neighbours of e inside the hit set, in the reference's element order, e itself excluded: record indices into out[], count
  // returned.  Same result as neighbours() below without building, sorting and de-duplicating a list of elements per record.
  int neighbour_records(const Elem &e, int *out, int cap) const
  {
    int cnt = 0;
    for (const auto &c : adj.cand[e.type]) {
      u64 ci = 0;
      bool ok = true;
      for (int d = 0; d < ND && ok; d ++) {
        const long long rel = (long long)e.c[d] + c.off[d] - lb[d];
        ok = rel >= 0 && rel < sz[d];
        ci += (u64)rel * prod[d];
      }
      const long long tt = (long long)e.c[ND] + c.off[ND];
      if (!ok || tt < 0) continue;
      ci += (u64)tt * prod[ND];
      const int r = find_tag(ci * (u64)NTYPES + (u64)c.type);
      if (r >= 0 && cnt < cap) out[cnt ++] = r;
    }
    return cnt;
  }
  // records that share a (d+1)-cell with e, in the element order of the reference's std::set, e itself excluded
  void neighbours(const Elem &e, std::vector<std::pair<Elem, int>> &out) const
  {
    out.clear();
    for (const auto &cell : adj.side_of[e.type]) {
      int cc[N];
      for (int a = 0; a < N; a ++) cc[a] = e.c[a] + cell.off[a];
      for (const auto &face : adj.sides[cell.type]) {
        Elem f; f.c[0] = f.c[1] = f.c[2] = f.c[3] = 0;
        for (int a = 0; a < N; a ++) f.c[a] = cc[a] + face.off[a];
        f.type = face.type;
        bool self = f.type == e.type;
        for (int a = 0; a < N && self; a ++) self = f.c[a] == e.c[a];
        if (self) continue;
        const int r = find(f);
        if (r >= 0) out.push_back({f, r});
      }
    }
    std::sort(out.begin(), out.end(), [](const std::pair<Elem, int> &a, const std::pair<Elem, int> &b) { return elem_less(a.first, b.first, N); });
    out.erase(std::unique(out.begin(), out.end(), [](const std::pair<Elem, int> &a, const std::pair<Elem, int> &b) { return a.second == b.second; }), out.end());
  }
};

struct UnionFind {
  std::vector<int> p;
  explicit UnionFind(size_t n) : p(n) { for (size_t i = 0; i < n; i ++) p[i] = (int)i; }
  int find(int x) { while (p[x] != x) { p[x] = p[p[x]]; x = p[x]; } return x; }
  void unite(int a, int b) { a = find(a); b = find(b); if (a != b) p[b] = a; }
};

// The same for several threads at once: a root is only ever linked under a root with a SMALLER index (compare-and-swap on the
// root's own parent entry), so parents decrease along every path -- no cycles, no locks; path halving is a benign race.  When all
// unions are in, the root of a component is its smallest member.
struct ConcurrentUnionFind {
  std::vector<int> p;
  explicit ConcurrentUnionFind(size_t n) : p(n) { for (size_t i = 0; i < n; i ++) p[i] = (int)i; }
  int find(int x)
  {
    for (;;) {
      const int px = __atomic_load_n(&p[x], __ATOMIC_RELAXED);
      if (px == x) return x;
      const int ppx = __atomic_load_n(&p[px], __ATOMIC_RELAXED);
      if (ppx != px) { int e = px; __atomic_compare_exchange_n(&p[x], &e, ppx, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED); }
      x = px;
    }
  }
  void unite(int a, int b)
  {
    for (;;) {
      a = find(a); b = find(b);
      if (a == b) return;
      if (a < b) std::swap(a, b);                              // a > b: a goes under b
      int e = a;
      if (__atomic_compare_exchange_n(&p[a], &e, b, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) return;
    }
  }
};

// runs f(begin, end) over [0, n) on up to `cap` host threads (the per-record work of pass 2 is independent: SURVEY 8 f2 -- "the
// sorted-tag binary searches are independent per hit").  The threads are kept: spawning 16 of them per call cost as much as the
// work they then did on a hit set of 60 000 records.  Chunks are handed out through an atomic counter; the caller works too.
class WorkerPool {
public:
  static WorkerPool &get()
  {
    // (a forked child inherits the object but not the threads: it gets a pool of its own; the parent's is leaked there)
    static WorkerPool *pool = nullptr;
    static pid_t owner = 0;
    static std::mutex mu;
    std::lock_guard<std::mutex> g(mu);
    if (!pool || owner != getpid()) { pool = new WorkerPool(); owner = getpid(); }
    return *pool;
  }
  void run(size_t n, unsigned want, const std::function<void(size_t, size_t)> &f)
  {
    std::lock_guard<std::mutex> serial(run_mu);               // one job at a time
    grow(want > 0 ? want - 1 : 0);
    const size_t nchunks = std::min<size_t>((size_t)want * 4, (n + 1023) / 1024);
    {
      std::lock_guard<std::mutex> g(mu);
      job = &f; job_n = n; job_chunks = nchunks; next.store(0); working = (unsigned)th.size(); gen ++;
    }
    cv.notify_all();
    work();
    std::unique_lock<std::mutex> g(mu);
    done_cv.wait(g, [&] { return working == 0; });
    job = nullptr;
  }
  ~WorkerPool()
  {
    { std::lock_guard<std::mutex> g(mu); stop = true; }
    cv.notify_all();
    for (auto &t : th) t.join();
  }

private:
  std::vector<std::thread> th;
  std::mutex mu, run_mu;
  std::condition_variable cv, done_cv;
  const std::function<void(size_t, size_t)> *job = nullptr;
  size_t job_n = 0, job_chunks = 0;
  std::atomic<size_t> next{0};
  unsigned working = 0;
  unsigned long long gen = 0;
  bool stop = false;

  void work()
  {
    for (;;) {
      const size_t c = next.fetch_add(1);
      if (c >= job_chunks) break;
      const size_t per = (job_n + job_chunks - 1) / job_chunks, b = std::min(job_n, c * per), e = std::min(job_n, b + per);
      if (b < e) (*job)(b, e);
    }
  }
  void grow(unsigned workers)
  {
    while (th.size() < workers) {
      const unsigned long long seen = gen;                     // a new worker waits for the NEXT job
      th.emplace_back([this, seen]() mutable {
        unsigned long long last = seen;
        for (;;) {
          std::unique_lock<std::mutex> g(mu);
          cv.wait(g, [&] { return stop || gen != last; });
          if (stop) return;
          last = gen;
          g.unlock();
          work();
          g.lock();
          if (-- working == 0) done_cv.notify_all();
        }
      });
    }
  }
};

template <class F>
void parallel_ranges(size_t n, F f, unsigned cap = 16, size_t min_n = 4096)
{
  unsigned nt = std::thread::hardware_concurrency();
  if (nt > cap) nt = cap;
  if (const char *e = getenv("FTKX_TRACE_THREADS")) nt = (unsigned)atoi(e);      // (an explicit choice is taken as it is)
  if (nt < 2 || n < min_n) { f((size_t)0, n); return; }
  const std::function<void(size_t, size_t)> fn = f;
  WorkerPool::get().run(n, nt, fn);
}

template <int N>
void Tracer<N>::build_hash(const ftkx_cp_t *recs, size_t n, bool parallel) { build_hash_tags(TagView(recs), n, parallel); }

template <int N>
template <class Tags>
void Tracer<N>::build_hash_tags(const Tags &tags, size_t n, bool parallel)
{
  size_t cap = 16;
  while (cap < 2 * n + 2) cap <<= 1;
  h_mask = cap - 1;
  h_slot.assign(cap, Slot{0, -1});
  auto insert = [&](size_t b, size_t e) {
    for (size_t i = b; i < e; i ++) {
      u64 p = mix(tags[i]) & h_mask;
      for (;;) {
        long long expect = -1;
        if (__atomic_compare_exchange_n(&h_slot[p].rec, &expect, (long long)i, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) break;
        p = (p + 1) & h_mask;
      }
      h_slot[p].tag = tags[i];
    }
  };
  if (parallel) parallel_ranges(n, insert); else insert(0, n);
}

// what the device has already done for a record set (trace_device.hip): the neighbours of every record inside the set, in the
// reference's element order (`maxnb` slots per record, `deg` of them filled), and the component root of every ordinary record
struct DevicePhase { const int *nbr; const unsigned char *deg; const int *root; int maxnb; };


template <int N>
int trace_impl(const long long *dst, const long long *dsz, const TagView tags, size_t n, ftkx_curves *out, const DevicePhase *dev = nullptr)
{
  static const Adjacency<N> adj;
  constexpr bool prof = false;      // (phase timing to stderr: a debugging aid, compiled out)
  auto now = [] { return std::chrono::duration<double>(std::chrono::high_resolution_clock::now().time_since_epoch()).count() * 1e3; };
  double tp[8]; int np_ = 0; tp[np_ ++] = now();
  constexpr int MAXNB = 2 * N;                       // two cells per face, N other faces each
  Tracer<N> tr(adj, dst, dsz);
  // duplicate tags are refused.  The sweep hands its records over sorted by tag: adjacent records are compared in place; only an
  // unsorted set pays for an index and a sort
  if (dev) { if (dev->maxnb != MAXNB) return FTKX_E_INVALID; }       // (the caller has checked the tags: strictly ascending)
  else {
    bool sorted = true;
    for (size_t i = 1; i < n && sorted; i ++) sorted = tags[i - 1] <= tags[i];
    if (sorted) { for (size_t i = 1; i < n; i ++) if (tags[i] == tags[i - 1]) return FTKX_E_INVALID; }
    else {
      tr.index.resize(n);
      for (size_t i = 0; i < n; i ++) tr.index[i] = {tags[i], (int)i};
      ftkx::sort_on_threads(tr.index);
      for (size_t i = 1; i < n; i ++) if (tr.index[i].first == tr.index[i - 1].first) return FTKX_E_INVALID;
    }
    tr.build_hash_tags(tags, n, true);
  }
  tp[np_ ++] = now();
  std::vector<Elem> elem(n);
  // adjacency lists inside the hit set (sorted in element order), flat: at most MAXNB neighbours per record
  std::vector<int> nbr_own(dev ? 0 : n * MAXNB);
  std::vector<unsigned char> deg_own(dev ? 0 : n);
  int *const nbr_w = nbr_own.data(); unsigned char *const deg_w = deg_own.data();
  const int *const nbr = dev ? dev->nbr : nbr_own.data();          // (the device's lists are used where they lie: pinned, cached host memory)
  const unsigned char *const deg = dev ? dev->deg : deg_own.data();
  // the reference's element order (corner as a vector with x FIRST, then the type) as one integer per record
  std::vector<std::pair<u64, int>> order_key(n);
  parallel_ranges(n, [&](size_t b, size_t e) {
    for (size_t i = b; i < e; i ++) {
      elem[i] = tr.decode(tags[i]);
      if (!dev) deg_w[i] = (unsigned char)tr.neighbour_records(elem[i], &nbr_w[i * MAXNB], MAXNB);
      u64 key = 0;
      for (int d = 0; d < ND_OF(N); d ++) key = key * (u64)dsz[d] + (u64)(elem[i].c[d] - dst[d]);
      key = (key << 24) | (u64)(unsigned)elem[i].c[N - 1];           // (time: below 2^24 steps, else the comparator sort below)
      order_key[i] = {key * (u64)Tracer<N>::NTYPES + (u64)elem[i].type, (int)i};
    }
  });
  tp[np_ ++] = now();
  struct NbRange { const int *b, *e; const int *begin() const { return b; } const int *end() const { return e; } size_t size() const { return (size_t)(e - b); } };
  auto nb = [&](size_t i) { return NbRange{&nbr[i * MAXNB], &nbr[i * MAXNB] + deg[i]}; };
  std::vector<char> ordinary(n);
  for (size_t i = 0; i < n; i ++) ordinary[i] = deg[i] <= 2;

  // curves = connected components of the ordinary nodes (threads: the unions of a range of records each, then everybody's root)
  std::vector<int> root_own(dev ? 0 : n);
  const int *const root = dev ? dev->root : root_own.data();
  if (!dev) {
    int *const root_w = root_own.data();
    ConcurrentUnionFind uf(n);
    parallel_ranges(n, [&](size_t b, size_t e) {
      for (size_t i = b; i < e; i ++)
        if (ordinary[i]) for (int j : nb(i)) if (ordinary[j] && j < (int)i) uf.unite((int)i, j);     // (every edge is seen from both ends: once is enough)
    });
    parallel_ranges(n, [&](size_t b, size_t e) { for (size_t i = b; i < e; i ++) root_w[i] = uf.find((int)i); });
  }
  // seeds: the smallest element of every component, components enumerated in element order of their seed (a sort of the seeds only)
  std::vector<int> order;
  bool key_ok = true;
  {
    long double span = 16777216.0L * (long double)Tracer<N>::NTYPES;
    for (int d = 0; d < ND_OF(N); d ++) span *= (long double)dsz[d];
    for (size_t i = 0; i < n && key_ok; i ++) key_ok = elem[i].c[N - 1] < (1 << 24);
    key_ok = key_ok && span < 18446744073709551615.0L;
  }
  if (key_ok) {
    // per root the smallest key of its members (atomic minimum), then the member that holds it (keys are unique: one writer)
    std::vector<u64> best_key(n, ~0ull);
    std::vector<int> best(n, -1);
    parallel_ranges(n, [&](size_t b, size_t e) {
      for (size_t i = b; i < e; i ++) {
        if (!ordinary[i]) continue;
        u64 *slot = &best_key[root[i]];
        const u64 k = order_key[i].first;
        u64 cur = __atomic_load_n(slot, __ATOMIC_RELAXED);
        while (k < cur && !__atomic_compare_exchange_n(slot, &cur, k, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
      }
    });
    parallel_ranges(n, [&](size_t b, size_t e) {
      for (size_t i = b; i < e; i ++) if (ordinary[i] && order_key[i].first == best_key[root[i]]) best[root[i]] = (int)i;
    });
    std::vector<std::pair<u64, int>> seeds;
    for (size_t r = 0; r < n; r ++) if (best[r] >= 0) seeds.push_back({order_key[best[r]].first, best[r]});
    std::sort(seeds.begin(), seeds.end());
    order.reserve(seeds.size());
    for (const auto &sd : seeds) order.push_back(sd.second);
  } else {
    order.resize(n);
    for (size_t i = 0; i < n; i ++) order[i] = (int)i;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return elem_less(elem[a], elem[b], N); });
  }
  tp[np_ ++] = now();

  std::vector<int> seq; seq.reserve(n);
  std::vector<long long> offsets(1, 0);
  std::vector<int> loops;
  std::vector<char> visited(n, 0);
  // one curve from its seed (cc2curves.hh:48-105): first towards the seed's smallest neighbour, then the other way; is_loop
  // (cc2curves.hh:113-122).  Touches `visited` of its own component only: curves can be walked side by side.
  auto walk = [&](int s, std::vector<int> &out_seq, std::vector<int> &front, std::vector<int> &back) -> int {
    front.clear(); back.clear();
    visited[s] = 1;
    int seed_nb[2 * N], nseed = 0;
    for (int j : nb(s)) if (ordinary[j]) seed_nb[nseed ++] = j;
    for (int dir = 0; dir < 2; dir ++) {
      if (nseed == 0) break;
      int cur = dir == 0 ? seed_nb[0] : seed_nb[nseed - 1];
      while (true) {
        if (!visited[cur]) { (dir == 0 ? back : front).push_back(cur); visited[cur] = 1; }
        int next = -1;
        for (int j : nb(cur)) if (ordinary[j] && !visited[j]) { next = j; break; }   // same component by construction
        if (next < 0) break;
        cur = next;
      }
      if (nseed == 1) break;
    }
    const size_t begin = out_seq.size();
    for (size_t i = front.size(); i > 0; i --) out_seq.push_back(front[i - 1]);
    out_seq.push_back(s);
    for (int v : back) out_seq.push_back(v);
    int loop = 0;
    if (out_seq.size() - begin >= 2) {
      const int f = out_seq[begin], l = out_seq.back();
      for (int j : nb(f)) if (j == l) loop = 1;
    }
    return loop;
  };
  if (key_ok && order.size() >= 8 && n >= 4096) {
    // `order` holds one seed per component: the curves are walked on the worker threads, each into its own list, and strung
    // together in seed order
    const size_t nc = order.size();
    std::vector<std::vector<int>> part(nc);
    std::vector<int> part_loop(nc, 0);
    std::atomic<size_t> next_curve{0};
    parallel_ranges(n, [&](size_t, size_t) {                   // (ranges ignored: the curves are handed out one by one)
      std::vector<int> front, back;
      for (;;) {
        const size_t c = next_curve.fetch_add(1);
        if (c >= nc) break;
        part_loop[c] = walk(order[c], part[c], front, back);
      }
    });
    for (size_t c = 0; c < nc; c ++) {
      seq.insert(seq.end(), part[c].begin(), part[c].end());
      offsets.push_back((long long)seq.size());
      loops.push_back(part_loop[c]);
    }
  } else {
    std::vector<char> seeded(n, 0);
    std::vector<int> front, back;
    for (int s : order) {
      if (!ordinary[s]) continue;
      if (seeded[root[s]]) continue;
      seeded[root[s]] = 1;
      const int loop = walk(s, seq, front, back);
      offsets.push_back((long long)seq.size());
      loops.push_back(loop);
    }
  }

  tp[np_ ++] = now();
  if (prof) fprintf(stderr, "trace: n %zu  index %.3f  neighbours %.3f  components+order %.3f  walk %.3f ms\n", n, tp[1] - tp[0], tp[2] - tp[1], tp[3] - tp[2], tp[4] - tp[3]);
  out->n_curves = loops.size();
  out->n_points = seq.size();
  out->offsets = (long long *)malloc(offsets.size() * sizeof(long long));
  out->indices = (long long *)malloc((seq.size() ? seq.size() : 1) * sizeof(long long));
  out->loop = (int *)malloc((loops.size() ? loops.size() : 1) * sizeof(int));
  if (!out->offsets || !out->indices || !out->loop) return FTKX_E_NOMEM;
  memcpy(out->offsets, offsets.data(), offsets.size() * sizeof(long long));
  for (size_t i = 0; i < seq.size(); i ++) out->indices[i] = seq[i];
  for (size_t i = 0; i < loops.size(); i ++) out->loop[i] = loops[i];
  size_t nspecial = 0;
  for (size_t i = 0; i < n; i ++) nspecial += !ordinary[i];
  out->n_special = nspecial;
  return FTKX_OK;
}

// trace_critical_points_online (filters/critical_point_tracker.hh:523-639), the enable_streaming_trajectories branch of
// update_timestep (2d:326-330, 3d:197-201): called after every interval sweep with the discrete points found since the last call.
//   1. every trajectory that is not complete is continued greedily, forwards from its last point and backwards from its first:
//      the FIRST neighbour (in the element order of the reference's std::set) that is among the new points is appended and
//      consumed, until none is left; a trajectory that could not be continued is complete;
//   2. the points left over form new trajectories: connected components -> cc2curves, exactly as offline (trace_impl);
//   3. the discrete points are forgotten.
template <int N>
int grow_impl(const long long *dst, const long long *dsz, std::vector<std::vector<ftkx_cp_t>> &curves, std::vector<int> &loop, std::vector<int> &complete,
              const ftkx_cp_t *recs, size_t n)
{
  static const Adjacency<N> adj;
  Tracer<N> tr(adj, dst, dsz);
  {
    std::vector<u64> tags(n);
    for (size_t i = 0; i < n; i ++) tags[i] = recs[i].tag;
    std::sort(tags.begin(), tags.end());
    for (size_t i = 1; i < n; i ++) if (tags[i] == tags[i - 1]) return FTKX_E_INVALID;
  }
  tr.build_hash(recs, n);
  std::vector<char> alive(n, 1);
  std::vector<std::pair<Elem, int>> tmp;
  for (size_t c = 0; c < curves.size(); c ++) {
    if (complete[c] || curves[c].empty()) continue;
    bool continued = false;
    for (int dir = 0; dir < 2; dir ++) {
      Elem cur = tr.decode(dir == 0 ? curves[c].back().tag : curves[c].front().tag);
      for (;;) {
        tr.neighbours(cur, tmp);
        int next = -1;
        for (const auto &cand : tmp) if (alive[cand.second]) { next = cand.second; cur = cand.first; break; }
        if (next < 0) break;
        if (dir == 0) curves[c].push_back(recs[next]); else curves[c].insert(curves[c].begin(), recs[next]);
        alive[next] = 0;
        continued = true;
      }
    }
    if (!continued) complete[c] = 1;
  }
  std::vector<ftkx_cp_t> rest;
  for (size_t i = 0; i < n; i ++) if (alive[i]) rest.push_back(recs[i]);
  const size_t nr = rest.size();
  ftkx_curves fresh;
  memset(&fresh, 0, sizeof(fresh));
  const int rc = trace_impl<N>(dst, dsz, rest.data(), nr, &fresh);
  if (rc != FTKX_OK) { free(fresh.offsets); free(fresh.indices); free(fresh.loop); return rc; }
  // The new trajectories are born in the order of the reference's extract_connected_components (algorithms/cca.hh:91-116): the
  // components come out of union_find::get_sets (basic/union_find.hh:121-133) keyed by their ROOT element, and which member is
  // the root follows from that union-find's union-by-size with its quirk -- uniting two members of one set (every element is
  // united with itself: neighbors() contains it) adds the root's size to itself.  Emulated literally, sizes modulo 2^64.
  Tracer<N> tr2(adj, dst, dsz);
  tr2.build_hash(rest.data(), nr);
  std::vector<Elem> el(nr);
  for (size_t i = 0; i < nr; i ++) el[i] = tr2.decode(rest[i].tag);
  std::vector<int> ord(nr);
  for (size_t i = 0; i < nr; i ++) ord[i] = (int)i;
  std::sort(ord.begin(), ord.end(), [&](int a, int b) { return elem_less(el[a], el[b], N); });
  std::vector<int> parent(nr);
  std::vector<u64> sz(nr, 1);
  for (size_t i = 0; i < nr; i ++) parent[i] = (int)i;
  auto find = [&](int x) { while (parent[x] != x) x = parent[x]; return x; };
  for (int cur : ord) {
    tr2.neighbours(el[cur], tmp);                         // in element order, without `cur` itself
    bool self_done = false;
    auto unite = [&](int a, int b) {
      a = find(a); b = find(b);
      if (sz[a] < sz[b]) { parent[a] = b; sz[b] += sz[a]; } else { parent[b] = a; sz[a] += sz[b]; }
    };
    for (const auto &cand : tmp) {
      if (!self_done && elem_less(el[cur], cand.first, N)) { unite(cur, cur); self_done = true; }
      unite(cur, cand.second);
    }
    if (!self_done) unite(cur, cur);
  }
  std::vector<std::pair<int, size_t>> birth;              // (rank of the component's root in element order, curve)
  {
    std::vector<int> rank(nr);
    for (size_t i = 0; i < nr; i ++) rank[ord[i]] = (int)i;
    for (size_t c = 0; c < fresh.n_curves; c ++)
      if (fresh.offsets[c + 1] > fresh.offsets[c]) birth.push_back({rank[find((int)fresh.indices[fresh.offsets[c]])], c});
    std::stable_sort(birth.begin(), birth.end(), [](const std::pair<int, size_t> &a, const std::pair<int, size_t> &b) { return a.first < b.first; });
  }
  for (const auto &bc : birth) {
    const size_t c = bc.second;
    std::vector<ftkx_cp_t> pts;
    for (long long k = fresh.offsets[c]; k < fresh.offsets[c + 1]; k ++) pts.push_back(rest[(size_t)fresh.indices[k]]);
    curves.push_back(std::move(pts)); loop.push_back(fresh.loop[c]); complete.push_back(0);
  }
  free(fresh.offsets); free(fresh.indices); free(fresh.loop);
  return FTKX_OK;
}

}  // namespace

struct ftkx_online_tracer {
  int nd;
  long long dst[3], dsz[3];
  std::vector<std::vector<ftkx_cp_t>> curves;
  std::vector<int> loop, complete;
};

namespace ftkx {
// the candidate tables of the neighbour search, flattened for the device: per face type the (type, offset[N]) of every record that can
// share a cell with it, in the reference's element order; cand_off has NTYPES + 1 entries; returns the slots per record (2 N)
int trace_candidates(int nd, std::vector<int> &cand_off, std::vector<int> &cand_flat)
{
  cand_off.clear(); cand_flat.clear();
  auto fill = [&](auto &adj, int N) {
    cand_off.push_back(0);
    for (const auto &v : adj.cand) {
      for (const auto &c : v) { cand_flat.push_back(c.type); for (int a = 0; a < N; a ++) cand_flat.push_back(c.off[a]); for (int a = N; a < 4; a ++) cand_flat.push_back(0); }
      cand_off.push_back((int)(cand_flat.size() / 5));
    }
  };
  if (nd == 2) { static const Adjacency<3> adj; fill(adj, 3); return 6; }
  static const Adjacency<4> adj; fill(adj, 4); return 8;
}

// ftkx_trace_curves with the neighbour search and the component labelling done elsewhere (tags strictly ascending)
int trace_curves_with(int nd, const long long *dst, const long long *dsz, const unsigned long long *tags, size_t n, ftkx_curves *out,
                      const int *nbr, const unsigned char *deg, const int *root, int maxnb)
{
  if ((nd != 2 && nd != 3) || !dst || !dsz || (!tags && n) || !out) return FTKX_E_INVALID;
  memset(out, 0, sizeof(*out));
  const DevicePhase dp{nbr, deg, root, maxnb};
  try { return nd == 2 ? trace_impl<3>(dst, dsz, TagView(tags), n, out, &dp) : trace_impl<4>(dst, dsz, TagView(tags), n, out, &dp); }
  catch (const std::bad_alloc &) { return FTKX_E_NOMEM; }
}

// (host only) the same on a tag array
int trace_curves_tags(int nd, const long long *dst, const long long *dsz, const unsigned long long *tags, size_t n, ftkx_curves *out)
{
  if ((nd != 2 && nd != 3) || !dst || !dsz || (!tags && n) || !out) return FTKX_E_INVALID;
  memset(out, 0, sizeof(*out));
  try { return nd == 2 ? trace_impl<3>(dst, dsz, TagView(tags), n, out) : trace_impl<4>(dst, dsz, TagView(tags), n, out); }
  catch (const std::bad_alloc &) { return FTKX_E_NOMEM; }
}
}  // namespace ftkx

extern "C" {

int ftkx_online_tracer_create(ftkx_online_tracer **out, int nd, const long long domain_st[3], const long long domain_sz[3])
{
  if (!out || (nd != 2 && nd != 3) || !domain_st || !domain_sz) return FTKX_E_INVALID;
  for (int d = 0; d < nd; d ++) if (domain_sz[d] <= 0) return FTKX_E_INVALID;
  ftkx_online_tracer *t = new ftkx_online_tracer();
  t->nd = nd;
  for (int d = 0; d < 3; d ++) { t->dst[d] = d < nd ? domain_st[d] : 0; t->dsz[d] = d < nd ? domain_sz[d] : 1; }
  *out = t;
  return FTKX_OK;
}

void ftkx_online_tracer_destroy(ftkx_online_tracer *t) { delete t; }

int ftkx_online_tracer_grow(ftkx_online_tracer *t, const ftkx_cp_t *recs, size_t n)
{
  if (!t || (n && !recs)) return FTKX_E_INVALID;
  return t->nd == 2 ? grow_impl<3>(t->dst, t->dsz, t->curves, t->loop, t->complete, recs, n)
                    : grow_impl<4>(t->dst, t->dsz, t->curves, t->loop, t->complete, recs, n);
}

int ftkx_online_tracer_curves(const ftkx_online_tracer *t, ftkx_cp_t **points, ftkx_curves *out)
{
  if (!t || !points || !out) return FTKX_E_INVALID;
  memset(out, 0, sizeof(*out));
  size_t np = 0;
  for (const auto &c : t->curves) np += c.size();
  *points = (ftkx_cp_t *)malloc((np ? np : 1) * sizeof(ftkx_cp_t));
  out->offsets = (long long *)malloc((t->curves.size() + 1) * sizeof(long long));
  out->indices = (long long *)malloc((np ? np : 1) * sizeof(long long));
  out->loop = (int *)malloc((t->curves.size() ? t->curves.size() : 1) * sizeof(int));
  if (!*points || !out->offsets || !out->indices || !out->loop) return FTKX_E_NOMEM;
  size_t k = 0;
  out->offsets[0] = 0;
  for (size_t c = 0; c < t->curves.size(); c ++) {
    for (const ftkx_cp_t &p : t->curves[c]) { (*points)[k] = p; out->indices[k] = (long long)k; k ++; }
    out->offsets[c + 1] = (long long)k;
    out->loop[c] = t->loop[c];
  }
  out->n_curves = t->curves.size(); out->n_points = np;
  return FTKX_OK;
}

int ftkx_trace_curves(int nd, const long long domain_st[3], const long long domain_sz[3], const ftkx_cp_t *recs, size_t n, ftkx_curves *out)
{
  if (!out || (n && !recs) || (nd != 2 && nd != 3)) return FTKX_E_INVALID;
  memset(out, 0, sizeof(*out));
  for (int d = 0; d < nd; d ++) if (domain_sz[d] <= 0) return FTKX_E_INVALID;
  return nd == 2 ? trace_impl<3>(domain_st, domain_sz, TagView(recs), n, out) : trace_impl<4>(domain_st, domain_sz, TagView(recs), n, out);
}

void ftkx_free_curves(ftkx_curves *c)
{
  if (!c) return;
  free(c->offsets); free(c->indices); free(c->loop);
  memset(c, 0, sizeof(*c));
}

// Trajectory post-processing with the defaults of json_interface::post_process (include/ftk/filters/json_interface.hh:758-800):
//   per curve   smooth_ordinal_types(2), smooth_interval_types(), rotate()          features/feature_curve.hh:295-348, 254-266
//   set         split_all(): curves of mixed type are cut into runs of one type      features/feature_curve_set.hh:514-532, feature_curve.hh:220-243
//   per curve   reorder() (ascending time), adjust_time() (monotone t)               features/feature_curve.hh:268-293
// Quirks kept: split() drops the first point of every new run (the point at which the type changes), split pieces are never
// loops, and a curve whose single type is 0 (UNKNOWN) counts as "inconsistent" and goes through split() too.
int ftkx_post_process_curves(const ftkx_cp_t *recs, size_t n, const ftkx_curves *in, ftkx_trajectories *out)
{
  if (!out || !in || (n && !recs)) return FTKX_E_INVALID;
  memset(out, 0, sizeof(*out));
  struct Pt { long long idx; unsigned type; double t; int ordinal, timestep; };
  struct Piece { size_t b, e; int loop, id; };               // a trajectory: points [b, e) of the flat array
  // Every traced curve is worked on by itself, in place in ONE flat array of points (smoothing, rotation; the pieces split_all cuts it
  // into are sub-ranges of its range; re-ordering and time adjustment work on those): no allocation per curve, and the curves are
  // handed to the host threads one by one (62 181 points in 72 curves of very different lengths: 2.9 ms on one thread with a vector
  // per curve, 0.9 ms on 16 threads with vectors, the allocator in everybody's way).
  const size_t nc = in->n_curves, np_in = in->n_points;
  for (size_t c = 0; c < nc; c ++) if (in->offsets[c] < 0 || in->offsets[c + 1] < in->offsets[c] || (size_t)in->offsets[c + 1] > np_in) return FTKX_E_INVALID;
  std::vector<Pt> P(np_in);
  std::vector<std::vector<Piece>> pieces(nc);
  std::atomic<int> bad{0};
  std::atomic<size_t> next_curve{0};
  auto work = [&](size_t, size_t) {                            // (ranges ignored: the curves are handed out one by one)
    std::vector<int> o;                                        // positions of the ordinal points of the curve in hand
    std::vector<std::pair<int, unsigned>> pending;
    for (;;) {
      const size_t c = next_curve.fetch_add(1);
      if (c >= nc) break;
      const size_t b0 = (size_t)in->offsets[c], e0 = (size_t)in->offsets[c + 1], len = e0 - b0;
      Pt *p = P.data() + b0;
      for (size_t k = 0; k < len; k ++) {
        const long long i = in->indices[b0 + k];
        if (i < 0 || (size_t)i >= n) { bad = 1; break; }
        p[k] = Pt{i, recs[i].type, recs[i].t, ftkx_cp_ordinal(&recs[i]), ftkx_cp_timestep(&recs[i])};
      }
      if (bad) break;
      const int loop = in->loop[c];
      auto ordinals = [&]() { o.clear(); for (size_t i = 0; i < len; i ++) if (p[i].ordinal) o.push_back((int)i); };
      {   // smooth_ordinal_types(half_window_size = 2)
        const int h = 2;
        ordinals();
        if ((int)o.size() >= 2 * h + 1) {
          pending.clear();
          for (int i = h; i < (int)o.size() - h; i ++) {
            unsigned consistent = p[o[i - h]].type;
            for (int j = i - h; j <= i + h; j ++) {
              if (j == i) continue;
              if (consistent != p[o[j]].type) { consistent = 0; break; }
            }
            if (consistent != 0 && p[o[i]].type != consistent) pending.push_back({o[i], consistent});
          }
          for (const auto &kv : pending) p[kv.first].type = kv.second;
        }
      }
      {   // smooth_interval_types (the ordinal positions are the same: types changed, places did not)
        if (!o.empty()) {
          const unsigned ft = p[o.front()].type;
          for (int i = 0; i < o.front(); i ++) p[i].type = ft;
          const unsigned bt = p[o.back()].type;
          for (int i = o.back(); i < (int)len; i ++) p[i].type = bt;
          for (size_t i = 0; i + 1 < o.size(); i ++) {
            if (p[o[i]].type == p[o[i + 1]].type) {
              const unsigned it = p[o[i]].type;
              for (int j = o[i]; j < o[i + 1]; j ++) p[j].type = it;
            } else {
              const unsigned lt = p[o[i]].type, rt = p[o[i + 1]].type;
              int j;
              for (j = o[i]; j < o[i + 1]; j ++) if (p[j].type != lt) break;
              for (; j < o[i + 1]; j ++) p[j].type = rt;
            }
          }
        }
      }
      // rotate
      if (loop && len && p[0].type == p[len - 1].type) {
        size_t i = 0;
        for (; i < len; i ++) if (p[0].type != p[i].type) break;
        if (i < len) std::rotate(p, p + i, p + len);
      }
      // split_all: runs of one type; the point that ends a run belongs to no piece (feature_curve.hh:220-243, reproduced)
      std::vector<Piece> &result = pieces[c];
      unsigned consistent = len ? p[0].type : 1u;
      for (size_t i = 0; i < len; i ++) if (p[i].type != consistent) { consistent = 0; break; }
      if (!len || consistent != 0) result.push_back(Piece{b0, e0, loop, (int)c});     // feature_curve_set_t::add numbers traced curves 0, 1, 2, ... (feature_curve_set.hh:458-465)
      else {
        size_t run_b = 0, run_n = 0;
        unsigned current = 0;
        for (size_t i = 0; i < len; i ++) {
          if (run_n == 0) { current = p[i].type; run_b = i; }
          if (p[i].type == current) run_n ++;
          if (p[i].type != current || i == len - 1) {
            if (run_n) { result.push_back(Piece{b0 + run_b, b0 + run_b + run_n, 0, (int)c}); run_n = 0; }   // split_all re-adds the pieces under their parent's label (feature_curve_set.hh:530-531)
          }
        }
      }
      for (const Piece &pc : result) {
        Pt *q = P.data() + pc.b;
        const size_t m = pc.e - pc.b;
        if (m && !pc.loop) {   // reorder
          bool reverse = false;
          if (q[0].timestep == q[m - 1].timestep) { if (q[0].t > q[m - 1].t) reverse = true; }
          else if (q[0].timestep > q[m - 1].timestep) reverse = true;
          if (reverse) std::reverse(q, q + m);
        }
        // adjust_time
        for (size_t i = 0; i < m; i ++) { if (i == 0 || q[i].ordinal) continue; q[i].t = std::max(q[i - 1].t, q[i].t); }
        for (size_t i = m; i -- > 0; ) { if (i == m - 1 || q[i].ordinal) continue; q[i].t = std::min(q[i + 1].t, q[i].t); }
      }
    }
  };
  parallel_ranges(np_in, work, 16, 8192);
  if (bad) return FTKX_E_INVALID;
  // the pieces strung together in the curves' order
  std::vector<size_t> first(nc + 1, 0);
  for (size_t c = 0; c < nc; c ++) first[c + 1] = first[c] + pieces[c].size();
  const size_t nres = first[nc];
  size_t np = 0;
  for (const auto &pc : pieces) for (const Piece &q : pc) np += q.e - q.b;
  out->n_curves = nres; out->n_points = np;
  out->offsets = (long long *)malloc((nres + 1) * sizeof(long long));
  out->indices = (long long *)malloc((np ? np : 1) * sizeof(long long));
  out->loop = (int *)malloc((nres ? nres : 1) * sizeof(int));
  out->type = (unsigned *)malloc((np ? np : 1) * sizeof(unsigned));
  out->t = (double *)malloc((np ? np : 1) * sizeof(double));
  out->id = (int *)malloc((nres ? nres : 1) * sizeof(int));
  if (!out->offsets || !out->indices || !out->loop || !out->type || !out->t || !out->id) return FTKX_E_NOMEM;
  out->offsets[0] = 0;
  for (size_t c = 0; c < nc; c ++)
    for (size_t j = 0; j < pieces[c].size(); j ++) out->offsets[first[c] + j + 1] = out->offsets[first[c] + j] + (long long)(pieces[c][j].e - pieces[c][j].b);
  std::atomic<size_t> next_out{0};
  parallel_ranges(np_in, [&](size_t, size_t) {
    for (;;) {
      const size_t c = next_out.fetch_add(1);
      if (c >= nc) break;
      for (size_t j = 0; j < pieces[c].size(); j ++) {
        const size_t r = first[c] + j;
        size_t k = (size_t)out->offsets[r];
        for (size_t i = pieces[c][j].b; i < pieces[c][j].e; i ++) { out->indices[k] = P[i].idx; out->type[k] = P[i].type; out->t[k] = P[i].t; k ++; }
        out->loop[r] = pieces[c][j].loop;
        out->id[r] = pieces[c][j].id;
      }
    }
  }, 16, 8192);
  return FTKX_OK;
}

void ftkx_free_trajectories(ftkx_trajectories *c)
{
  if (!c) return;
  free(c->offsets); free(c->indices); free(c->loop); free(c->type); free(c->t); free(c->id);
  memset(c, 0, sizeof(*c));
}

}  // extern "C"
