// Record-stream formats of the reference, written and read on the host.
//
// Discrete (untraced) critical points -- what `ftk -f cp --output-type discrete` produces
//   JSON    critical_point_tracker.hh:106-108 (get_critical_points_json), read back at :498-508; one object per point with
//           the members of features/feature_point.hh:143-171 (nlohmann adl_serializer: x, t, timestep, scalar, v, type,
//           ordinal, tag, id); nlohmann objects are std::map, so the members come out in alphabetical order.
//   binary  critical_point_tracker.hh:339-352: diy::serializeToFile(std::vector<feature_point_t>) = size_t count, then per
//           point the fields of features/feature_point.hh:175-205 back to back (105 bytes, no padding).
//   text    critical_point_tracker.hh:392-396 -> feature_point_t::print (features/feature_point.hh:86-105), ostream defaults
//           (6 significant digits).
// Traced trajectories -- `--output-type traced`
//   JSON    {"trajs": [curve, ...]}, features/feature_curve_set.hh:75-90; curve = features/feature_curve.hh:436-467
//   binary  features/feature_curve_set.hh:94-118 (count; per curve: int label, curve), curve = features/feature_curve.hh:472-510
//   text    features/feature_curve_set.hh:180-228, including its unclosed "bbmin=(x, y, z, " parentheses
// Per-curve statistics are feature_curve_t::update_statistics (features/feature_curve.hh:145-180).
//
// The JSON number spelling follows nlohmann 3.11.2 (the version vendored by the reference): shortest round-trip digits,
// plain notation for decimal exponents in (-4, 15], "d.ddde-07" otherwise, integral values with a trailing ".0", non-finite
// values as null.  Files written here and by the reference parse to the same values; the binary files are byte-identical.
#include <algorithm>
#include <charconv>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/ftkx.h"
#include "internal.hpp"

namespace {

typedef unsigned long long u64;

struct Point {   // == ftk::feature_point_t (features/feature_point.hh:128-139)
  double x[3] = {0, 0, 0}, t = 0;
  int timestep = 0;
  double scalar[3] = {0, 0, 0}, v[3] = {0, 0, 0};
  unsigned type = 0;
  bool ordinal = false;
  u64 tag = 0, id = 0;
};

struct Curve {   // == ftk::feature_curve_t minus what no format stores (features/feature_curve.hh:45-52)
  int id = 0;
  bool complete = false, loop = false;
  double max[3], min[3], persistence[3], bbmin[3], bbmax[3], tmin, tmax;
  unsigned consistent_type = 0;
  std::vector<Point> pts;
};

int io_fail(int code, const char *fmt, ...)
{
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  ftkx::set_global_error(buf);
  return code;
}

Point point_of(const ftkx_cp_t &r, const double *v, const u64 *id, size_t i)
{
  Point p;
  for (int k = 0; k < 3; k ++) { p.x[k] = r.x[k]; p.scalar[k] = r.scalar[k]; if (v) p.v[k] = v[3 * i + k]; }
  p.t = r.t; p.type = r.type; p.tag = r.tag;
  p.ordinal = ftkx_cp_ordinal(&r) != 0; p.timestep = ftkx_cp_timestep(&r);
  if (id) p.id = id[i];
  return p;
}

void record_of(const Point &p, ftkx_cp_t *r)
{
  memset(r, 0, sizeof(*r));
  for (int k = 0; k < 3; k ++) { r->x[k] = p.x[k]; r->scalar[k] = p.scalar[k]; }
  r->t = p.t; r->type = p.type; r->tag = p.tag;
  ((unsigned int *)r)[15] = ((unsigned)p.timestep << 1) | (p.ordinal ? 1u : 0u);
}

// feature_curve_t::update_statistics (features/feature_curve.hh:145-180); std::max(a, b) = (a < b) ? b : a keeps `a` on NaN
void update_statistics(Curve &c)
{
  if (c.pts.empty()) return;
  const double lo = std::numeric_limits<double>::lowest(), hi = std::numeric_limits<double>::max();
  for (int k = 0; k < 3; k ++) { c.max[k] = lo; c.min[k] = hi; c.bbmax[k] = lo; c.bbmin[k] = hi; }
  c.tmax = lo; c.tmin = hi;
  for (const Point &p : c.pts) {
    for (int k = 0; k < 3; k ++) {
      c.max[k] = std::max(c.max[k], p.scalar[k]); c.min[k] = std::min(c.min[k], p.scalar[k]);
      c.bbmax[k] = std::max(c.bbmax[k], p.x[k]); c.bbmin[k] = std::min(c.bbmin[k], p.x[k]);
    }
    c.tmax = std::max(c.tmax, p.t); c.tmin = std::min(c.tmin, p.t);
  }
  for (int k = 0; k < 3; k ++) c.persistence[k] = c.max[k] - c.min[k];
  c.consistent_type = c.pts[0].type;
  for (const Point &p : c.pts) if (p.type != c.consistent_type) { c.consistent_type = 0; break; }
}

// ------------------------------------------------------------------------------------------------------------------
// JSON out
// ------------------------------------------------------------------------------------------------------------------
void json_double(std::string &o, double v)
{
  if (!std::isfinite(v)) { o += "null"; return; }
  if (v == 0) { o += std::signbit(v) ? "-0.0" : "0.0"; return; }
  char sci[40];
  const auto res = std::to_chars(sci, sci + sizeof(sci) - 1, v, std::chars_format::scientific);   // shortest round-trip: d[.ddd]e[+-]XX
  *res.ptr = 0;
  const char *p = sci, *end = res.ptr;
  if (*p == '-') { o += '-'; p ++; }
  char digits[24];
  int k = 0;
  const char *e = p;
  for (; e < end && *e != 'e'; e ++) if (*e != '.') digits[k ++] = *e;
  const int exp10 = atoi(e + 1);
  const int n = exp10 + 1;                         // position of the decimal point relative to the first digit
  if (k <= n && n <= 15) { o.append(digits, k); o.append((size_t)(n - k), '0'); o += ".0"; return; }
  if (0 < n && n <= 15) { o.append(digits, n); o += '.'; o.append(digits + n, k - n); return; }
  if (-4 < n && n <= 0) { o += "0."; o.append((size_t)(-n), '0'); o.append(digits, k); return; }
  o += digits[0];
  if (k > 1) { o += '.'; o.append(digits + 1, k - 1); }
  o += 'e';
  int ex = n - 1;
  if (ex < 0) { o += '-'; ex = -ex; } else o += '+';
  char eb[8];
  snprintf(eb, sizeof(eb), "%02d", ex);
  o += eb;
}

void json_array3(std::string &o, const double a[3])
{
  o += '[';
  for (int k = 0; k < 3; k ++) { if (k) o += ','; json_double(o, a[k]); }
  o += ']';
}

void json_u64(std::string &o, u64 v) { char b[24]; snprintf(b, sizeof(b), "%llu", v); o += b; }
void json_int(std::string &o, long long v) { char b[24]; snprintf(b, sizeof(b), "%lld", v); o += b; }

void json_point(std::string &o, const Point &p)
{
  o += "{\"id\":"; json_u64(o, p.id);
  o += ",\"ordinal\":"; o += p.ordinal ? "true" : "false";
  o += ",\"scalar\":"; json_array3(o, p.scalar);
  o += ",\"t\":"; json_double(o, p.t);
  o += ",\"tag\":"; json_u64(o, p.tag);
  o += ",\"timestep\":"; json_int(o, p.timestep);
  o += ",\"type\":"; json_u64(o, p.type);
  o += ",\"v\":"; json_array3(o, p.v);
  o += ",\"x\":"; json_array3(o, p.x);
  o += '}';
}

void json_curve(std::string &o, const Curve &c)
{
  o += "{\"bbmax\":"; json_array3(o, c.bbmax);
  o += ",\"bbmin\":"; json_array3(o, c.bbmin);
  o += ",\"consistent_type\":"; json_u64(o, c.consistent_type);
  o += ",\"id\":"; json_int(o, c.id);
  o += ",\"max\":"; json_array3(o, c.max);
  o += ",\"min\":"; json_array3(o, c.min);
  o += ",\"persistence\":"; json_array3(o, c.persistence);
  o += ",\"tmax\":"; json_double(o, c.tmax);
  o += ",\"tmin\":"; json_double(o, c.tmin);
  o += ",\"traj\":[";
  for (size_t i = 0; i < c.pts.size(); i ++) { if (i) o += ','; json_point(o, c.pts[i]); }
  o += "]}";
}

// ------------------------------------------------------------------------------------------------------------------
// JSON in: a small recursive-descent reader (objects, arrays, numbers, strings, true/false/null)
// ------------------------------------------------------------------------------------------------------------------
struct JValue {
  enum Kind { NUL, BOOL, NUM, STR, ARR, OBJ } kind = NUL;
  bool b = false;
  double num = 0;
  u64 unum = 0;            // exact value of a non-negative integer literal (tags exceed 2^53)
  long long inum = 0;
  bool is_int = false;
  std::string str;
  std::vector<JValue> arr;
  std::vector<std::pair<std::string, JValue>> obj;
  const JValue *get(const char *key) const
  {
    for (const auto &kv : obj) if (kv.first == key) return &kv.second;
    return nullptr;
  }
};

struct JParser {
  const char *p, *end;
  std::string err;
  void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) p ++; }
  bool fail(const char *m) { if (err.empty()) err = m; return false; }
  bool lit(const char *s) { const size_t n = strlen(s); if ((size_t)(end - p) < n || memcmp(p, s, n)) return fail("bad literal"); p += n; return true; }
  bool string(std::string &out)
  {
    if (p >= end || *p != '"') return fail("expected string");
    p ++;
    while (p < end && *p != '"') {
      if (*p == '\\') {
        if (++ p >= end) return fail("bad escape");
        switch (*p) {
          case 'n': out += '\n'; break; case 't': out += '\t'; break; case 'r': out += '\r'; break;
          case 'b': out += '\b'; break; case 'f': out += '\f'; break;
          case 'u': { if (end - p < 5) return fail("bad \\u"); unsigned cp = (unsigned)strtoul(std::string(p + 1, p + 5).c_str(), nullptr, 16); out += (char)(cp < 128 ? cp : '?'); p += 4; break; }
          default: out += *p;
        }
        p ++;
      } else out += *p ++;
    }
    if (p >= end) return fail("unterminated string");
    p ++;
    return true;
  }
  bool value(JValue &v, int depth = 0)
  {
    if (depth > 64) return fail("nesting too deep");
    ws();
    if (p >= end) return fail("unexpected end");
    if (*p == '{') {
      v.kind = JValue::OBJ; p ++; ws();
      if (p < end && *p == '}') { p ++; return true; }
      for (;;) {
        ws();
        std::string key;
        if (!string(key)) return false;
        ws();
        if (p >= end || *p != ':') return fail("expected ':'");
        p ++;
        v.obj.emplace_back(std::move(key), JValue());
        if (!value(v.obj.back().second, depth + 1)) return false;
        ws();
        if (p < end && *p == ',') { p ++; continue; }
        if (p < end && *p == '}') { p ++; return true; }
        return fail("expected ',' or '}'");
      }
    }
    if (*p == '[') {
      v.kind = JValue::ARR; p ++; ws();
      if (p < end && *p == ']') { p ++; return true; }
      for (;;) {
        v.arr.emplace_back();
        if (!value(v.arr.back(), depth + 1)) return false;
        ws();
        if (p < end && *p == ',') { p ++; continue; }
        if (p < end && *p == ']') { p ++; return true; }
        return fail("expected ',' or ']'");
      }
    }
    if (*p == '"') { v.kind = JValue::STR; return string(v.str); }
    if (*p == 't') { v.kind = JValue::BOOL; v.b = true; return lit("true"); }
    if (*p == 'f') { v.kind = JValue::BOOL; v.b = false; return lit("false"); }
    if (*p == 'n') { v.kind = JValue::NUL; return lit("null"); }
    // number
    const char *s = p;
    bool integral = true;
    if (p < end && *p == '-') p ++;
    while (p < end && ((*p >= '0' && *p <= '9') || *p == '.' || *p == 'e' || *p == 'E' || *p == '+' || *p == '-')) {
      if (*p == '.' || *p == 'e' || *p == 'E') integral = false;
      p ++;
    }
    if (p == s) return fail("unexpected character");
    const std::string tok(s, p);
    v.kind = JValue::NUM;
    v.num = strtod(tok.c_str(), nullptr);
    if (integral) {
      v.is_int = true;
      if (tok[0] == '-') { v.inum = strtoll(tok.c_str(), nullptr, 10); v.unum = (u64)v.inum; }
      else { v.unum = strtoull(tok.c_str(), nullptr, 10); v.inum = (long long)v.unum; }
    }
    return true;
  }
};

double j_double(const JValue *v) { return (v && v->kind == JValue::NUM) ? v->num : std::numeric_limits<double>::quiet_NaN(); }   // null (non-finite on write) -> NaN
u64 j_u64(const JValue *v) { return (v && v->kind == JValue::NUM) ? (v->is_int ? v->unum : (u64)v->num) : 0; }
long long j_int(const JValue *v) { return (v && v->kind == JValue::NUM) ? (v->is_int ? v->inum : (long long)v->num) : 0; }
bool j_bool(const JValue *v) { return v && ((v->kind == JValue::BOOL && v->b) || (v->kind == JValue::NUM && v->num != 0)); }
bool j_array3(const JValue *v, double out[3])
{
  if (!v || v->kind != JValue::ARR || v->arr.size() != 3) return false;
  for (int k = 0; k < 3; k ++) out[k] = j_double(&v->arr[k]);
  return true;
}

bool point_from_json(const JValue &j, Point &p)   // adl_serializer<feature_point_t>::from_json, features/feature_point.hh:158-170
{
  if (j.kind != JValue::OBJ) return false;
  if (!j_array3(j.get("x"), p.x) || !j_array3(j.get("scalar"), p.scalar) || !j_array3(j.get("v"), p.v)) return false;
  if (!j.get("t") || !j.get("timestep") || !j.get("type") || !j.get("ordinal") || !j.get("tag") || !j.get("id")) return false;
  p.t = j_double(j.get("t"));
  p.timestep = (int)j_int(j.get("timestep"));
  p.type = (unsigned)j_u64(j.get("type"));
  p.ordinal = j_bool(j.get("ordinal"));
  p.tag = j_u64(j.get("tag"));
  p.id = j_u64(j.get("id"));
  return true;
}

// ------------------------------------------------------------------------------------------------------------------
// binary (diy::save of PODs: raw bytes in declaration order)
// ------------------------------------------------------------------------------------------------------------------
struct BinOut {
  std::string s;
  template <class T> void put(const T &v) { s.append(reinterpret_cast<const char *>(&v), sizeof(T)); }
  void point(const Point &p)
  {
    put(p.x); put(p.t); put(p.timestep); put(p.scalar); put(p.v); put(p.type);
    const unsigned char o = p.ordinal ? 1 : 0; put(o);
    put(p.tag); put(p.id);
  }
};
constexpr size_t kPointBytes = 24 + 8 + 4 + 24 + 24 + 4 + 1 + 8 + 8;   // 105

struct BinIn {
  const char *p, *end;
  bool ok = true;
  template <class T> void get(T &v) { if ((size_t)(end - p) < sizeof(T)) { ok = false; memset(&v, 0, sizeof(T)); return; } memcpy(&v, p, sizeof(T)); p += sizeof(T); }
  void point(Point &q)
  {
    get(q.x); get(q.t); get(q.timestep); get(q.scalar); get(q.v); get(q.type);
    unsigned char o = 0; get(o); q.ordinal = o != 0;
    get(q.tag); get(q.id);
  }
};

// ------------------------------------------------------------------------------------------------------------------
// text (std::ostream << double at default precision == "%g")
// ------------------------------------------------------------------------------------------------------------------
void text_double(std::string &o, double v) { char b[40]; snprintf(b, sizeof(b), "%g", v); o += b; }

void text_point(std::string &o, const Point &p, const std::vector<std::string> &names)   // feature_point_t::print
{
  o += "x=("; text_double(o, p.x[0]); o += ", "; text_double(o, p.x[1]); o += ", "; text_double(o, p.x[2]); o += "), ";
  o += "t="; text_double(o, p.t); o += ", ";
  for (size_t k = 0; k < names.size(); k ++) { o += names[k]; o += '='; text_double(o, p.scalar[k]); o += ", "; }
  o += "v=("; text_double(o, p.v[0]); o += ", "; text_double(o, p.v[1]); o += ", "; text_double(o, p.v[2]); o += "), ";
  o += "type="; json_u64(o, p.type); o += ", ";
  o += "timestep="; json_int(o, p.timestep); o += ", ";
  o += "ordinal="; o += p.ordinal ? '1' : '0'; o += ", ";
  o += "tag="; json_u64(o, p.tag); o += ", ";
  o += "id="; json_u64(o, p.id);
}

std::vector<std::string> names_of(const char *const *names, int n)
{
  std::vector<std::string> out;
  if (n < 0) { out.push_back("scalar"); return out; }   // the tracker's default scalar_components (critical_point_tracker.hh:183)
  for (int i = 0; i < n && i < 3; i ++) out.push_back(names && names[i] ? names[i] : "scalar");
  return out;
}

int write_file(const char *path, const std::string &s)
{
  FILE *fp = fopen(path, "wb");
  if (!fp) return io_fail(FTKX_E_INVALID, "cannot open %s for writing", path);
  const size_t w = s.empty() ? 0 : fwrite(s.data(), 1, s.size(), fp);
  const int rc = fclose(fp);
  if (w != s.size() || rc != 0) return io_fail(FTKX_E_INVALID, "short write to %s", path);
  return FTKX_OK;
}

int read_file(const char *path, std::string &s)
{
  FILE *fp = fopen(path, "rb");
  if (!fp) return io_fail(FTKX_E_INVALID, "cannot open %s", path);
  char buf[1 << 16];
  size_t n;
  while ((n = fread(buf, 1, sizeof(buf), fp)) > 0) s.append(buf, n);
  fclose(fp);
  return FTKX_OK;
}

int curves_of(const ftkx_cp_t *recs, size_t n, const ftkx_trajectories *tr, std::vector<Curve> &out)
{
  if (!tr || (n && !recs)) return io_fail(FTKX_E_INVALID, "traced writer: null argument");
  if (tr->n_curves && (!tr->offsets || !tr->loop)) return io_fail(FTKX_E_INVALID, "traced writer: incomplete ftkx_trajectories");
  out.resize(tr->n_curves);
  for (size_t c = 0; c < tr->n_curves; c ++) {
    Curve &cv = out[c];
    cv.id = tr->id ? tr->id[c] : (int)c;
    cv.loop = tr->loop[c] != 0;
    if (tr->offsets[c] < 0 || tr->offsets[c + 1] < tr->offsets[c] || (size_t)tr->offsets[c + 1] > tr->n_points)
      return io_fail(FTKX_E_INVALID, "traced writer: bad offsets");
    for (long long k = tr->offsets[c]; k < tr->offsets[c + 1]; k ++) {
      const long long i = tr->indices[k];
      if (i < 0 || (size_t)i >= n) return io_fail(FTKX_E_INVALID, "traced writer: point index %lld out of range", i);
      Point p = point_of(recs[i], nullptr, nullptr, 0);
      if (tr->type) p.type = tr->type[k];
      if (tr->t) p.t = tr->t[k];
      p.id = (u64)(long long)cv.id;                 // feature_curve_t::relabel (features/feature_curve.hh:99-104)
      cv.pts.push_back(p);
    }
    if (cv.pts.empty()) {                           // update_statistics leaves an empty curve untouched; keep the file deterministic
      for (int k = 0; k < 3; k ++) cv.max[k] = cv.min[k] = cv.persistence[k] = cv.bbmin[k] = cv.bbmax[k] = 0;
      cv.tmin = cv.tmax = 0;
    }
    update_statistics(cv);
  }
  return FTKX_OK;
}

int trajectories_of(const std::vector<Curve> &curves, ftkx_cp_t **recs, size_t *n, ftkx_trajectories *tr)
{
  size_t np = 0;
  for (const Curve &c : curves) np += c.pts.size();
  const size_t nc = curves.size();
  ftkx_cp_t *r = (ftkx_cp_t *)malloc((np ? np : 1) * sizeof(ftkx_cp_t));
  memset(tr, 0, sizeof(*tr));
  tr->offsets = (long long *)malloc((nc + 1) * sizeof(long long));
  tr->indices = (long long *)malloc((np ? np : 1) * sizeof(long long));
  tr->loop = (int *)malloc((nc ? nc : 1) * sizeof(int));
  tr->id = (int *)malloc((nc ? nc : 1) * sizeof(int));
  tr->type = (unsigned *)malloc((np ? np : 1) * sizeof(unsigned));
  tr->t = (double *)malloc((np ? np : 1) * sizeof(double));
  if (!r || !tr->offsets || !tr->indices || !tr->loop || !tr->id || !tr->type || !tr->t) {
    free(r); ftkx_free_trajectories(tr);
    return io_fail(FTKX_E_NOMEM, "out of memory");
  }
  size_t k = 0;
  tr->offsets[0] = 0;
  for (size_t c = 0; c < nc; c ++) {
    for (const Point &p : curves[c].pts) { record_of(p, &r[k]); tr->indices[k] = (long long)k; tr->type[k] = p.type; tr->t[k] = p.t; k ++; }
    tr->offsets[c + 1] = (long long)k;
    tr->loop[c] = curves[c].loop; tr->id[c] = curves[c].id;
  }
  tr->n_curves = nc; tr->n_points = np;
  *recs = r; *n = np;
  return FTKX_OK;
}

}  // namespace

extern "C" {

int ftkx_format_from_path(const char *path)   // filters/json_interface.hh:225-231 (vtp is not produced here)
{
  if (!path) return FTKX_FORMAT_BINARY;
  const size_t n = strlen(path);
  auto ends = [&](const char *e) { const size_t m = strlen(e); return n >= m && memcmp(path + n - m, e, m) == 0; };
  if (ends("txt")) return FTKX_FORMAT_TEXT;
  if (ends("json")) return FTKX_FORMAT_JSON;
  return FTKX_FORMAT_BINARY;
}

int ftkx_write_critical_points(const char *path, int format, const ftkx_cp_t *recs, size_t n, const double *v, const unsigned long long *id,
                               const char *const *scalar_names, int n_scalar_names)
{
  if (!path || (n && !recs)) return io_fail(FTKX_E_INVALID, "ftkx_write_critical_points: null argument");
  std::string s;
  if (format == FTKX_FORMAT_JSON) {
    s.reserve(n * 170 + 2);
    s += '[';
    for (size_t i = 0; i < n; i ++) { if (i) s += ','; json_point(s, point_of(recs[i], v, id, i)); }
    s += ']';
  } else if (format == FTKX_FORMAT_BINARY) {
    BinOut b;
    b.s.reserve(8 + n * kPointBytes);
    b.put((size_t)n);
    for (size_t i = 0; i < n; i ++) b.point(point_of(recs[i], v, id, i));
    s.swap(b.s);
  } else if (format == FTKX_FORMAT_TEXT) {
    const std::vector<std::string> names = names_of(scalar_names, n_scalar_names);
    for (size_t i = 0; i < n; i ++) { text_point(s, point_of(recs[i], v, id, i), names); s += '\n'; }
  } else return io_fail(FTKX_E_INVALID, "ftkx_write_critical_points: unknown format %d", format);
  return write_file(path, s);
}

int ftkx_read_critical_points(const char *path, int format, ftkx_cp_t **recs, size_t *n, double **v, unsigned long long **id)
{
  if (!path || !recs || !n) return io_fail(FTKX_E_INVALID, "ftkx_read_critical_points: null argument");
  *recs = nullptr; *n = 0;
  if (v) *v = nullptr;
  if (id) *id = nullptr;
  std::string s;
  if (int rc = read_file(path, s)) return rc;
  std::vector<Point> pts;
  if (format == FTKX_FORMAT_JSON) {
    JParser jp{s.data(), s.data() + s.size(), std::string()};
    JValue root;
    if (!jp.value(root)) return io_fail(FTKX_E_INVALID, "%s: JSON error: %s", path, jp.err.c_str());
    if (root.kind != JValue::ARR) return io_fail(FTKX_E_INVALID, "%s: expected an array of points", path);
    pts.resize(root.arr.size());
    for (size_t i = 0; i < pts.size(); i ++)
      if (!point_from_json(root.arr[i], pts[i])) return io_fail(FTKX_E_INVALID, "%s: point %zu lacks a member", path, i);
  } else if (format == FTKX_FORMAT_BINARY) {
    BinIn b{s.data(), s.data() + s.size()};
    size_t cnt = 0;
    b.get(cnt);
    if (!b.ok || cnt > (s.size() - 8) / kPointBytes) return io_fail(FTKX_E_INVALID, "%s: truncated (count %zu, %zu bytes)", path, cnt, s.size());
    pts.resize(cnt);
    for (size_t i = 0; i < cnt; i ++) b.point(pts[i]);
  } else return io_fail(FTKX_E_UNSUPPORTED, "ftkx_read_critical_points: the reference has no reader for format %d", format);
  const size_t cnt = pts.size();
  ftkx_cp_t *r = (ftkx_cp_t *)malloc((cnt ? cnt : 1) * sizeof(ftkx_cp_t));
  double *vv = v ? (double *)malloc((cnt ? cnt : 1) * 3 * sizeof(double)) : nullptr;
  unsigned long long *ii = id ? (unsigned long long *)malloc((cnt ? cnt : 1) * sizeof(unsigned long long)) : nullptr;
  if (!r || (v && !vv) || (id && !ii)) { free(r); free(vv); free(ii); return io_fail(FTKX_E_NOMEM, "out of memory"); }
  for (size_t i = 0; i < cnt; i ++) {
    record_of(pts[i], &r[i]);
    if (vv) for (int k = 0; k < 3; k ++) vv[3 * i + k] = pts[i].v[k];
    if (ii) ii[i] = pts[i].id;
  }
  *recs = r; *n = cnt;
  if (v) *v = vv;
  if (id) *id = ii;
  return FTKX_OK;
}

int ftkx_write_traced_critical_points(const char *path, int format, const ftkx_cp_t *recs, size_t n, const ftkx_trajectories *trajs,
                                      const char *const *scalar_names, int n_scalar_names)
{
  if (!path) return io_fail(FTKX_E_INVALID, "ftkx_write_traced_critical_points: null path");
  std::vector<Curve> curves;
  if (int rc = curves_of(recs, n, trajs, curves)) return rc;
  std::string s;
  if (format == FTKX_FORMAT_JSON) {
    s += "{\"trajs\":[";
    for (size_t c = 0; c < curves.size(); c ++) { if (c) s += ','; json_curve(s, curves[c]); }
    s += "]}";
  } else if (format == FTKX_FORMAT_BINARY) {
    BinOut b;
    b.put((size_t)curves.size());
    for (const Curve &c : curves) {
      b.put(c.id);
      const unsigned char complete = c.complete ? 1 : 0; b.put(complete);
      b.put(c.max); b.put(c.min); b.put(c.persistence); b.put(c.bbmin); b.put(c.bbmax); b.put(c.tmin); b.put(c.tmax);
      b.put(c.consistent_type);
      b.put((size_t)c.pts.size());
      for (const Point &p : c.pts) b.point(p);
    }
    s.swap(b.s);
  } else if (format == FTKX_FORMAT_TEXT) {
    const std::vector<std::string> names = names_of(scalar_names, n_scalar_names);
    s += "#trajectories="; json_u64(s, curves.size()); s += '\n';
    for (const Curve &c : curves) {
      s += "--trajectory "; json_int(s, c.id); s += ", ";
      if (!names.empty()) {
        const struct { const char *label; const double *a; } groups[3] = {{"min=(", c.min}, {"max=(", c.max}, {"persistence=(", c.persistence}};
        for (const auto &g : groups) {
          s += g.label;
          for (size_t k = 0; k < names.size(); k ++) { text_double(s, g.a[k]); s += (k + 1 < names.size()) ? ", " : "), "; }
        }
      }
      s += "bbmin=("; for (int k = 0; k < 3; k ++) { text_double(s, c.bbmin[k]); s += ", "; }   // never closed in the reference
      s += "bbmax=("; for (int k = 0; k < 3; k ++) { text_double(s, c.bbmax[k]); s += ", "; }
      s += "tmin="; text_double(s, c.tmin); s += ", tmax="; text_double(s, c.tmax); s += ", ";
      s += "consistent_type="; json_u64(s, c.consistent_type); s += ", ";
      s += "loop="; s += c.loop ? '1' : '0'; s += '\n';
      for (const Point &p : c.pts) { s += "---"; text_point(s, p, names); s += '\n'; }
    }
  } else return io_fail(FTKX_E_INVALID, "ftkx_write_traced_critical_points: unknown format %d", format);
  return write_file(path, s);
}

int ftkx_read_traced_critical_points(const char *path, int format, ftkx_cp_t **recs, size_t *n, ftkx_trajectories *trajs)
{
  if (!path || !recs || !n || !trajs) return io_fail(FTKX_E_INVALID, "ftkx_read_traced_critical_points: null argument");
  *recs = nullptr; *n = 0;
  memset(trajs, 0, sizeof(*trajs));
  std::string s;
  if (int rc = read_file(path, s)) return rc;
  std::vector<Curve> curves;
  if (format == FTKX_FORMAT_JSON) {
    JParser jp{s.data(), s.data() + s.size(), std::string()};
    JValue root;
    if (!jp.value(root)) return io_fail(FTKX_E_INVALID, "%s: JSON error: %s", path, jp.err.c_str());
    const JValue *list = root.get("trajs");
    if (root.kind != JValue::OBJ || !list || list->kind != JValue::ARR) return io_fail(FTKX_E_INVALID, "%s: expected {\"trajs\": [...]}", path);
    curves.resize(list->arr.size());
    for (size_t c = 0; c < curves.size(); c ++) {
      const JValue &jc = list->arr[c];
      const JValue *traj = jc.get("traj");
      if (jc.kind != JValue::OBJ || !traj || traj->kind != JValue::ARR) return io_fail(FTKX_E_INVALID, "%s: curve %zu has no traj", path, c);
      Curve &cv = curves[c];
      cv.id = (int)c;                              // feature_curve_set_t::from_list numbers the curves afresh (feature_curve_set.hh:130-134)
      cv.pts.resize(traj->arr.size());
      for (size_t i = 0; i < cv.pts.size(); i ++)
        if (!point_from_json(traj->arr[i], cv.pts[i])) return io_fail(FTKX_E_INVALID, "%s: curve %zu point %zu lacks a member", path, c, i);
    }
  } else if (format == FTKX_FORMAT_BINARY) {
    BinIn b{s.data(), s.data() + s.size()};
    size_t nc = 0;
    b.get(nc);
    if (!b.ok || nc > s.size()) return io_fail(FTKX_E_INVALID, "%s: truncated", path);
    curves.resize(nc);
    for (size_t c = 0; c < nc; c ++) {
      Curve &cv = curves[c];
      unsigned char complete = 0;
      size_t np = 0;
      b.get(cv.id); b.get(complete); cv.complete = complete != 0;
      b.get(cv.max); b.get(cv.min); b.get(cv.persistence); b.get(cv.bbmin); b.get(cv.bbmax); b.get(cv.tmin); b.get(cv.tmax);
      b.get(cv.consistent_type); b.get(np);
      if (!b.ok || np > (size_t)(b.end - b.p) / kPointBytes) return io_fail(FTKX_E_INVALID, "%s: truncated in curve %zu", path, c);
      cv.pts.resize(np);
      for (size_t i = 0; i < np; i ++) { b.point(cv.pts[i]); cv.pts[i].id = (u64)(long long)cv.id; }   // relabel(id) on load
    }
  } else return io_fail(FTKX_E_UNSUPPORTED, "ftkx_read_traced_critical_points: the reference has no reader for format %d", format);
  return trajectories_of(curves, recs, n, trajs);
}

}  // extern "C"
