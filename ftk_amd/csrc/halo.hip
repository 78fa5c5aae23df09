// Compact t-slab halo: a slab boundary slice handed to the neighbouring rank as sign masks + patches of input values
// around the cells that survive the neighbour's cull, instead of the slice itself.  No reference counterpart (the reference cuts
// space, not time: include/ftk/filters/regular_tracker.hh:127-149).
#include "ctx.hpp"

using namespace ftkxh;

extern "C" {

// ---- compact t-slab halo ------------------------------------------------------------------------------------------------------
static int copy_out(ftkx_ctx *c, void *dst, const void *src, size_t bytes, int dst_on_device)
{
  if (!bytes) return FTKX_OK;
  HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, dst_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->stream));
  return FTKX_OK;
}

int ftkx_export_masks_size(ftkx_ctx *c, int t, size_t *u_bytes_out, size_t *n_words, unsigned long long *mask_factor, double *max_abs)
{
  if (c) c->ahead.clear();
  if (!c) return fail(nullptr, FTKX_E_INVALID, "null context");
  auto it = c->slices.find(t);
  if (it == c->slices.end()) return fail(c, FTKX_E_NOSLICE, "ftkx_export_masks_size: timestep %d not resident", t);
  Slice &s = it->second;
  if (!s.M || !s.U || !s.mask_factor || !s.max_known())
    return fail(c, FTKX_E_UNSUPPORTED, "ftkx_export_masks_size: slice %d has no summarised masks (ftkx_slices_prepare first; needs a mesh the two-level cull supports)", t);
  HIP_TRY(c, hipSetDevice(c->device));
  Mesh m; fill_mesh(c, m);
  for (int attempt = 0; attempt < 2; attempt ++) {
    HIP_TRY(c, hipMemsetAsync(c->d_counters + ftkx::CNT_SPARSE, 0, sizeof(u64), c->stream));
    ftkx::launch_compact_words(m, s.U, s.M, c->d_word_idx, c->d_words, c->words_cap, c->d_counters + ftkx::CNT_SPARSE, c->stream);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(c->h_counters, c->d_counters + ftkx::CNT_SPARSE, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const size_t n = (size_t)c->h_counters[0];
    if (n <= c->words_cap) { c->n_words = n; c->words_t = t; break; }
    if (c->d_word_idx) (void)hipFree(c->d_word_idx);
    if (c->d_words) (void)hipFree(c->d_words);
    c->d_word_idx = nullptr; c->d_words = nullptr; c->words_cap = 0;
    const size_t cap = n + n / 8 + 1024;
    HIP_TRY(c, hipMalloc((void **)&c->d_word_idx, cap * sizeof(unsigned)));
    HIP_TRY(c, hipMalloc((void **)&c->d_words, cap * sizeof(u64)));
    c->words_cap = cap;
  }
  if (u_bytes_out) *u_bytes_out = u_bytes_used(c, m);
  if (n_words) *n_words = c->n_words;
  if (mask_factor) *mask_factor = s.mask_factor;
  if (max_abs) *max_abs = s.maxabs;
  return FTKX_OK;
}

int ftkx_export_masks(ftkx_ctx *c, int t, void *U_dst, unsigned *word_index_dst, unsigned long long *words_dst, int dst_on_device)
{
  if (!c || !U_dst) return fail(c, FTKX_E_INVALID, "null argument");
  auto it = c->slices.find(t);
  if (it == c->slices.end() || c->words_t != t) return fail(c, FTKX_E_INVALID, "ftkx_export_masks: call ftkx_export_masks_size for timestep %d first", t);
  HIP_TRY(c, hipSetDevice(c->device));
  int rc;
  Mesh m; fill_mesh(c, m);
  if ((rc = copy_out(c, U_dst, it->second.U, u_bytes_used(c, m), dst_on_device))) return rc;
  if (c->n_words && (!word_index_dst || !words_dst)) return fail(c, FTKX_E_INVALID, "ftkx_export_masks: null list buffers");
  if ((rc = copy_out(c, word_index_dst, c->d_word_idx, c->n_words * sizeof(unsigned), dst_on_device))) return rc;
  if ((rc = copy_out(c, words_dst, c->d_words, c->n_words * sizeof(u64), dst_on_device))) return rc;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return FTKX_OK;
}

namespace {
u64 *halo_bad_flag(ftkx_ctx *c) { return c->d_counters + ftkx::CNT_N + 128; }
size_t pad8(size_t v) { return (v + 7) / 8 * 8; }
int factor_log2(unsigned long long f) { int b = 0; while (b < 63 && (1ull << b) < f) b ++; return b; }
// capacity of the word list in a packed mask message: the mask kernels write a word only where its summary is 0 -- a thin shell around
// the zero sets of the components; 1/64 of all words is generous for smooth data, and a slice that needs more is sent as it is
size_t packed_word_capacity(const ftkx_ctx *c) { return std::max<size_t>(4096, mask_bytes(c) / 8 / 64); }
}

}  // extern "C"

namespace ftkxh {
// layout of a packed mask message for this context's mesh: 32-byte header | summary array | word indices | words
bool packed_layout(const ftkx_ctx *c, const Mesh &m, size_t *ub, size_t *cap, size_t *off_idx, size_t *off_words, size_t *total)
{
  if (!ftkx::masks_have_summary(m)) return false;
  *ub = u_bytes_used(c, m); *cap = packed_word_capacity(c);
  *off_idx = 32 + pad8(*ub); *off_words = *off_idx + pad8(*cap * sizeof(unsigned)); *total = *off_words + *cap * sizeof(u64);
  return true;
}

// a slice that exists as masks only (the halo of a t-slab partition): zeroed field array (only patches are ever read), mask and summary
// arrays; an existing masks-only slice at t is kept as it is, a full slice there is an error (the caller drops it first)
int ensure_sparse_slice(ftkx_ctx *c, int t, int scalar_input)
{
  if (c->slices.empty()) c->scalar_mode = -1;
  if (c->scalar_mode >= 0 && c->scalar_mode != (scalar_input ? 1 : 0)) return fail(c, FTKX_E_INVALID, "halo slice: scalar and vector slices cannot be mixed in one context");
  auto it = c->slices.find(t);
  if (it != c->slices.end()) return it->second.sparse ? FTKX_OK : fail(c, FTKX_E_INVALID, "halo slice: timestep %d is resident as a full slice (drop it first)", t);
  c->scalar_mode = scalar_input ? 1 : 0;
  Slice s;
  const size_t n = n_vertices(c), ncomp = scalar_input ? 1 : (size_t)c->nd;
  auto fill = [&]() -> int {
    int rc;
    double **field = scalar_input ? &s.S : &s.V;
    HIP_TRY(c, hipMalloc((void **)field, n * ncomp * sizeof(double)));
    (scalar_input ? s.ownS : s.ownV) = true;
    HIP_TRY(c, hipMemsetAsync(*field, 0, n * ncomp * sizeof(double), c->stream));
    if ((rc = ensure_mask_arrays(c, s, true))) return rc;
    return FTKX_OK;
  };
  const int rc = fill();
  if (rc != FTKX_OK) { free_slice(s, c); return rc; }
  s.sparse = true;
  s.mask_gen = ++ c->mask_epoch;
  c->slices[t] = s;
  return FTKX_OK;
}
}  // namespace ftkxh

extern "C" {

int ftkx_push_masked_slice(ftkx_ctx *c, int t, int scalar_input, const void *U, size_t u_bytes_given, const unsigned *word_index, const unsigned long long *words, size_t n_words,
                           unsigned long long mask_factor, double max_abs, int on_device)
{
  if (c) c->ahead.clear();
  if (!c || !U) return fail(c, FTKX_E_INVALID, "null argument");
  if (!c->mesh_set) return fail(c, FTKX_E_INVALID, "push: call ftkx_set_mesh first");
  if (!c->pending.empty()) return fail(c, FTKX_E_INVALID, "push: sweeps pending, collect first");
  if (t < 0 || !pow2_factor(mask_factor) || !(max_abs >= 0)) return fail(c, FTKX_E_INVALID, "ftkx_push_masked_slice: bad arguments");
  if (c->slices.empty()) c->scalar_mode = -1;
  if (c->scalar_mode >= 0 && c->scalar_mode != (scalar_input ? 1 : 0)) return fail(c, FTKX_E_INVALID, "push: scalar and vector slices cannot be mixed in one context");
  HIP_TRY(c, hipSetDevice(c->device));
  const int saved_mode = c->scalar_mode;
  c->scalar_mode = scalar_input ? 1 : 0;
  Mesh m; fill_mesh(c, m);
  if (!ftkx::masks_have_summary(m)) { c->scalar_mode = saved_mode; return fail(c, FTKX_E_UNSUPPORTED, "ftkx_push_masked_slice: this mesh has no summarised masks"); }
  // (the summary array's size follows from the mesh AND the mask settings -- rows per summary byte: a sender configured differently
  // must not be read past its buffer, nor be taken for what it is not)
  if (u_bytes_given != u_bytes_used(c, m)) { c->scalar_mode = saved_mode; return fail(c, FTKX_E_INVALID, "ftkx_push_masked_slice: %zu summary bytes given, this mesh has %zu (different extents or FTKX_MASK_* settings on the sender?)", u_bytes_given, u_bytes_used(c, m)); }
  if (n_words > mask_bytes(c) / 8) { c->scalar_mode = saved_mode; return fail(c, FTKX_E_INVALID, "ftkx_push_masked_slice: more mask words (%zu) than the slice has", n_words); }
  auto it = c->slices.find(t);
  Slice s;
  if (it != c->slices.end() && it->second.sparse) { s = it->second; c->slices.erase(it); }          // the same halo slice again: keep its arrays
  else if (it != c->slices.end()) { free_slice(it->second, c); c->slices.erase(it); }
  const size_t n = n_vertices(c), ncomp = scalar_input ? 1 : (size_t)c->nd;
  // everything below that can fail runs inside `fill`: on failure the half-built slice is released, not leaked
  auto fill = [&]() -> int {
  int rc;
  if (!s.sparse) {
    double **field = scalar_input ? &s.S : &s.V;
    HIP_TRY(c, hipMalloc((void **)field, n * ncomp * sizeof(double)));
    (scalar_input ? s.ownS : s.ownV) = true;
    HIP_TRY(c, hipMemsetAsync(*field, 0, n * ncomp * sizeof(double), c->stream));      // only patches are ever read; zeros elsewhere, not garbage
    if ((rc = ensure_mask_arrays(c, s, true))) return rc;
    s.sparse = true;
  }
  const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  HIP_TRY(c, hipMemcpyAsync(s.U, U, u_bytes_used(c, m), kind, c->stream));
  if (n_words) {
    if (c->words_cap < n_words) {
      if (c->d_word_idx) (void)hipFree(c->d_word_idx);
      if (c->d_words) (void)hipFree(c->d_words);
      c->d_word_idx = nullptr; c->d_words = nullptr; c->words_cap = 0;
      HIP_TRY(c, hipMalloc((void **)&c->d_word_idx, n_words * sizeof(unsigned)));
      HIP_TRY(c, hipMalloc((void **)&c->d_words, n_words * sizeof(u64)));
      c->words_cap = n_words;
    }
    c->words_t = -1;
    HIP_TRY(c, hipMemcpyAsync(c->d_word_idx, word_index, n_words * sizeof(unsigned), kind, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_words, words, n_words * sizeof(u64), kind, c->stream));
    HIP_TRY(c, hipMemsetAsync(halo_bad_flag(c), 0, sizeof(u64), c->stream));
    ftkx::launch_scatter_words(c->d_word_idx, c->d_words, n_words, s.M, mask_bytes(c) / 8, halo_bad_flag(c), c->stream);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(c->h_counters, halo_bad_flag(c), sizeof(u64), hipMemcpyDeviceToHost, c->stream));
  }
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (n_words && c->h_counters[0]) return fail(c, FTKX_E_INVALID, "ftkx_push_masked_slice: word indices outside the mask array (a sender with another mesh?)");
  return FTKX_OK;
  };
  const int frc = fill();
  if (frc != FTKX_OK) { free_slice(s, c); c->scalar_mode = saved_mode; return frc; }
  s.mask_factor = mask_factor; s.mask_big = false; s.u_rows = m.u_rows;
  s.maxabs = max_abs;                                                                  // (all a masked slice knows of its values)
  s.mask_gen = ++ c->mask_epoch;
  c->slices[t] = s;
  return FTKX_OK;
}

// ---- the same hand-over as ONE message whose numbers stay on the device ------------------------------------------------------------
size_t ftkx_packed_masks_bytes(const ftkx_ctx *c, size_t *word_capacity)
{
  if (!c || !c->mesh_set) return 0;
  Mesh m; fill_mesh(c, m);
  if (!ftkx::masks_have_summary(m)) return 0;
  const size_t cap = packed_word_capacity(c);
  if (word_capacity) *word_capacity = cap;
  return 32 + pad8(u_bytes_used(c, m)) + pad8(cap * sizeof(unsigned)) + cap * sizeof(u64);
}

int ftkx_export_masks_packed(ftkx_ctx *c, int t, void *dst, int dst_on_device)
{
  if (c) c->ahead.clear();
  if (!c || !dst) return fail(c, FTKX_E_INVALID, "null argument");
  auto it = c->slices.find(t);
  if (it == c->slices.end()) return fail(c, FTKX_E_NOSLICE, "ftkx_export_masks_packed: timestep %d not resident", t);
  Slice &s = it->second;
  if (!s.M || !s.U || !s.mask_factor || !s.max_known())
    return fail(c, FTKX_E_UNSUPPORTED, "ftkx_export_masks_packed: slice %d has no summarised masks (ftkx_slices_prepare first; needs a mesh the two-level cull supports)", t);
  HIP_TRY(c, hipSetDevice(c->device));
  Mesh m; fill_mesh(c, m);
  size_t cap = 0;
  const size_t total = ftkx_packed_masks_bytes(c, &cap), ub = u_bytes_used(c, m);
  char *out = (char *)dst;
  if (!dst_on_device) {                                    // host-side callers (gloo): build it in device memory, copy out
    if (c->packed_cap < total) { if (c->d_packed) (void)hipFree(c->d_packed); c->d_packed = nullptr; c->packed_cap = 0; HIP_TRY(c, hipMalloc(&c->d_packed, total)); c->packed_cap = total; }
    out = (char *)c->d_packed;
  }
  unsigned *idx = (unsigned *)(out + 32 + pad8(ub));
  u64 *words = (u64 *)(out + 32 + pad8(ub) + pad8(cap * sizeof(unsigned)));
  HIP_TRY(c, hipMemsetAsync(c->d_counters + ftkx::CNT_SPARSE, 0, sizeof(u64), c->stream));
  ftkx::launch_compact_words(m, s.U, s.M, idx, words, cap, c->d_counters + ftkx::CNT_SPARSE, c->stream);
  // (the header says under which factor the masks were built and how many rows a summary byte stands for: the receiver checks both)
  ftkx::launch_pack_masks((u64 *)out, c->d_counters + ftkx::CNT_SPARSE, s.U, ub, cap, s.u_rows, factor_log2(s.mask_factor), c->stream);
  HIP_TRY(c, hipGetLastError());
  if (!dst_on_device) { HIP_TRY(c, hipMemcpyAsync(dst, out, total, hipMemcpyDeviceToHost, c->stream)); HIP_TRY(c, hipStreamSynchronize(c->stream)); }
  return FTKX_OK;                                          // (device destination: queued on the context's stream, nothing waited for)
}

int ftkx_push_masked_slice_packed(ftkx_ctx *c, int t, int scalar_input, const void *src, int src_on_device, unsigned long long mask_factor, double max_abs)
{
  if (c) c->ahead.clear();
  if (!c || !src) return fail(c, FTKX_E_INVALID, "null argument");
  if (!c->mesh_set) return fail(c, FTKX_E_INVALID, "push: call ftkx_set_mesh first");
  if (!c->pending.empty()) return fail(c, FTKX_E_INVALID, "push: sweeps pending, collect first");
  if (t < 0 || !pow2_factor(mask_factor) || !(max_abs >= 0)) return fail(c, FTKX_E_INVALID, "ftkx_push_masked_slice_packed: bad arguments");
  if (c->slices.empty()) c->scalar_mode = -1;
  if (c->scalar_mode >= 0 && c->scalar_mode != (scalar_input ? 1 : 0)) return fail(c, FTKX_E_INVALID, "push: scalar and vector slices cannot be mixed in one context");
  HIP_TRY(c, hipSetDevice(c->device));
  const int saved_mode = c->scalar_mode;
  c->scalar_mode = scalar_input ? 1 : 0;
  Mesh m; fill_mesh(c, m);
  size_t cap = 0;
  const size_t total = ftkx_packed_masks_bytes(c, &cap), ub = u_bytes_used(c, m);
  if (!total) { c->scalar_mode = saved_mode; return fail(c, FTKX_E_UNSUPPORTED, "ftkx_push_masked_slice_packed: this mesh has no summarised masks"); }
  auto it = c->slices.find(t);
  Slice s;
  if (it != c->slices.end() && it->second.sparse) { s = it->second; c->slices.erase(it); }          // the same halo slice again: keep its arrays
  else if (it != c->slices.end()) { free_slice(it->second, c); c->slices.erase(it); }
  const size_t n = n_vertices(c), ncomp = scalar_input ? 1 : (size_t)c->nd;
  auto fill = [&]() -> int {
    int rc;
    if (!s.sparse) {
      double **field = scalar_input ? &s.S : &s.V;
      HIP_TRY(c, hipMalloc((void **)field, n * ncomp * sizeof(double)));
      (scalar_input ? s.ownS : s.ownV) = true;
      HIP_TRY(c, hipMemsetAsync(*field, 0, n * ncomp * sizeof(double), c->stream));      // only patches are ever read; zeros elsewhere, not garbage
      if ((rc = ensure_mask_arrays(c, s, true))) return rc;
      s.sparse = true;
    }
    const char *in = (const char *)src;
    if (!src_on_device) {
      if (c->packed_cap < total) { if (c->d_packed) (void)hipFree(c->d_packed); c->d_packed = nullptr; c->packed_cap = 0; HIP_TRY(c, hipMalloc(&c->d_packed, total)); c->packed_cap = total; }
      HIP_TRY(c, hipMemcpyAsync(c->d_packed, src, total, hipMemcpyHostToDevice, c->stream));
      in = (const char *)c->d_packed;
    }
    // summaries and words: count, geometry, the sender's mask settings and factor (it must not exceed the factor the receiver was told:
    // masks serve their own factor and larger ones) and every index are checked on the device; a message that does not fit raises the
    // flag ftkx_sweep_cull looks at
    ftkx::launch_scatter_packed((const u64 *)in, (const unsigned *)(in + 32 + pad8(ub)), (const u64 *)(in + 32 + pad8(ub) + pad8(cap * sizeof(unsigned))), ub, cap, m.u_rows,
                                factor_log2(mask_factor), s.U, s.M, mask_bytes(c) / 8, halo_bad_flag(c), c->stream);
    HIP_TRY(c, hipGetLastError());
    if (!src_on_device) HIP_TRY(c, hipStreamSynchronize(c->stream));
    return FTKX_OK;
  };
  const int frc = fill();
  if (frc != FTKX_OK) { free_slice(s, c); c->scalar_mode = saved_mode; return frc; }
  s.mask_factor = mask_factor; s.mask_big = false; s.u_rows = m.u_rows;
  s.maxabs = max_abs;
  s.mask_gen = ++ c->mask_epoch;
  c->slices[t] = s;
  return FTKX_OK;
}

int ftkx_sweep_cull(ftkx_ctx *c, int t_sparse, size_t *n_cells)
{
  if (c) c->ahead.clear();
  if (!c || !n_cells) return fail(c, FTKX_E_INVALID, "null argument");
  *n_cells = 0;
  auto it = c->slices.find(t_sparse);
  if (it == c->slices.end() || !it->second.sparse) return fail(c, FTKX_E_INVALID, "ftkx_sweep_cull: timestep %d is not a masked halo slice", t_sparse);
  if (c->pending.empty()) return FTKX_OK;
  for (const Request &r : c->pending) if (r.mode != MODE_FAST) return fail(c, FTKX_E_UNSUPPORTED, "ftkx_sweep_cull: the pending sweeps do not use the cull (send the slice itself)");
  HIP_TRY(c, hipSetDevice(c->device));
  const double *field = it->second.S ? it->second.S : it->second.V;
  int rc;
  if ((rc = ensure_hit_buffer(c, std::max<u64>(c->capacity, 1u << 16)))) return rc;
  if ((rc = ensure_list(c, std::max<u64>(c->list_capacity, 1u << 20))) || (rc = ensure_refine(c, std::max<u64>(c->refine_capacity, 1u << 20)))) return rc;
  for (int attempt = 0; attempt < 4; attempt ++) {
    if (c->cells_cap < c->list_capacity) {
      if (c->d_cells) (void)hipFree(c->d_cells);
      c->d_cells = nullptr; c->cells_cap = 0;
      HIP_TRY(c, hipMalloc((void **)&c->d_cells, c->list_capacity * sizeof(u64)));
      c->cells_cap = c->list_capacity;
    }
    HIP_TRY(c, hipMemsetAsync(c->d_counters, 0, ftkx::CNT_N * sizeof(u64), c->stream));
    if ((rc = run_batch(c, field))) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->h_counters, c->d_counters, ftkx::CNT_N * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    u64 halo_bad = 0;
    HIP_TRY(c, hipMemcpyAsync(&halo_bad, halo_bad_flag(c), sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (halo_bad) {      // the packed mask message of this slice did not fit (more words than it holds, or another geometry): its masks are incomplete
      HIP_TRY(c, hipMemsetAsync(halo_bad_flag(c), 0, sizeof(u64), c->stream));
      return fail(c, FTKX_E_NOSLICE, "ftkx_sweep_cull: the packed masks of halo slice %d were incomplete (send the slice itself)", t_sparse);
    }
    for (auto &e : c->events) { ev_give(c, e.second.first); ev_give(c, e.second.second); }
    c->events.clear();
    const u64 listed = c->h_counters[ftkx::CNT_SURVIVOR_LIST], refined = std::max(c->h_counters[ftkx::CNT_REFINE_LIST], c->h_counters[ftkx::CNT_REFINE_PEAK]);
    if (listed <= c->list_capacity && refined <= c->refine_capacity) { c->n_cells = (size_t)c->h_counters[ftkx::CNT_SPARSE]; *n_cells = c->n_cells; return FTKX_OK; }
    if (refined > c->refine_capacity && (rc = ensure_refine(c, refined + refined / 8 + 1024))) return rc;
    if (listed > c->list_capacity && (rc = ensure_list(c, 2 * listed + 1024))) return rc;
  }
  return fail(c, FTKX_E_DEVICE, "ftkx_sweep_cull: survivor lists kept overflowing");
}

int ftkx_get_sparse_cells(ftkx_ctx *c, unsigned long long *dst, int dst_on_device)
{
  if (!c || (c->n_cells && !dst)) return fail(c, FTKX_E_INVALID, "null argument");
  HIP_TRY(c, hipSetDevice(c->device));
  int rc = copy_out(c, dst, c->d_cells, c->n_cells * sizeof(u64), dst_on_device);
  if (rc) return rc;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return FTKX_OK;
}

size_t ftkx_patch_doubles(const ftkx_ctx *c) { return c ? (size_t)(c->nd == 3 ? 216 : 36) * (size_t)(c->scalar_mode == 1 ? 1 : c->nd) : 0; }

static int patches_common(ftkx_ctx *c, int t, const unsigned long long *cells, size_t n, double *patches, int on_device, bool scatter)
{
  if (!c || (n && (!cells || !patches))) return fail(c, FTKX_E_INVALID, "null argument");
  auto it = c->slices.find(t);
  if (it == c->slices.end()) return fail(c, FTKX_E_NOSLICE, "patches: timestep %d not resident", t);
  if (scatter && !it->second.sparse) return fail(c, FTKX_E_INVALID, "ftkx_scatter_patches: timestep %d is not a masked halo slice", t);
  if (!n) return FTKX_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  Mesh m; fill_mesh(c, m);
  const int ncomp = c->scalar_mode == 1 ? 1 : c->nd;
  double *field = c->scalar_mode == 1 ? it->second.S : it->second.V;
  const size_t pd = ftkx_patch_doubles(c);
  const u64 *d_cells = cells; double *d_patches = patches;
  if (!on_device) {                              // host-side callers (gloo tests): stage through device buffers
    if (c->patch_cap < n) {
      if (c->d_patch_cells) (void)hipFree(c->d_patch_cells);
      if (c->d_patches) (void)hipFree(c->d_patches);
      c->d_patch_cells = nullptr; c->d_patches = nullptr; c->patch_cap = 0;
      HIP_TRY(c, hipMalloc((void **)&c->d_patch_cells, n * sizeof(u64)));
      HIP_TRY(c, hipMalloc((void **)&c->d_patches, n * pd * sizeof(double)));
      c->patch_cap = n;
    }
    HIP_TRY(c, hipMemcpyAsync(c->d_patch_cells, cells, n * sizeof(u64), hipMemcpyHostToDevice, c->stream));
    if (scatter) HIP_TRY(c, hipMemcpyAsync(c->d_patches, patches, n * pd * sizeof(double), hipMemcpyHostToDevice, c->stream));
    d_cells = c->d_patch_cells; d_patches = c->d_patches;
  }
  ftkx::launch_patches(m, scatter, d_cells, n, ncomp, field, d_patches, c->stream);
  HIP_TRY(c, hipGetLastError());
  if (!on_device && !scatter) HIP_TRY(c, hipMemcpyAsync(patches, c->d_patches, n * pd * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return FTKX_OK;
}

int ftkx_gather_patches(ftkx_ctx *c, int t, const unsigned long long *cells, size_t n, double *patches, int on_device)
{ return patches_common(c, t, cells, n, patches, on_device, false); }
int ftkx_scatter_patches(ftkx_ctx *c, int t, const unsigned long long *cells, size_t n, const double *patches, int on_device)
{ return patches_common(c, t, cells, n, const_cast<double *>(patches), on_device, true); }

}  // extern "C"
