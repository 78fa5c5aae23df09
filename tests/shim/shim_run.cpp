// Runs include/ftkx_shim.hh (instantiated with this repo's own lattice / record types) on fields read from a file and writes the
// records to another: tests/test_shim.py compares them with the Python binding of the same C ABI.
//   shim_run <in> <out>      in: int32 nd, nv(unused), DW, DH, DD, t, scope; u64 factor; then Vc, Vn, Jc, Jn, Sc, Sn as float64 arrays
#include <cstdint>
#include <cstdio>
#include <vector>

#include <ftkx_shim.hh>
#include <ftkx_tracker.hh>

static std::vector<double> rd(FILE *fp, size_t n) { std::vector<double> v(n); if (n && fread(v.data(), 8, n, fp) != n) { perror("read"); exit(2); } return v; }

int main(int argc, char **argv)
{
  if (argc < 3) return 2;
  FILE *fp = fopen(argv[1], "rb");
  if (!fp) { perror(argv[1]); return 2; }
  int32_t h[7];
  uint64_t factor;
  if (fread(h, 4, 7, fp) != 7 || fread(&factor, 8, 1, fp) != 1) return 2;
  const int nd = h[0], DW = h[2], DH = h[3], DD = h[4], t = h[5], scope = h[6];
  const size_t nv = (size_t)DW * DH * (nd == 3 ? DD : 1);
  const auto Vc = rd(fp, nv * nd), Vn = rd(fp, nv * nd), Jc = rd(fp, nv * nd * nd), Jn = rd(fp, nv * nd * nd), Sc = rd(fp, nv), Sn = rd(fp, nv);
  fclose(fp);
  std::vector<ftkx_cp_t> recs;
  try {
    if (nd == 2) {
      const ftkx::lattice dom({2, 2, 0}, {DW - 3, DH - 3, 2147483647LL}), core({2, 2, t}, {DW - 3, DH - 3, 1}), ext({0, 0}, {DW, DH});
      recs = ftkx::extract_cp2dt_hip<ftkx_cp_t>(scope, t, dom, core, ext, Vc.data(), Vn.data(), Jc.data(), Jn.data(), Sc.data(), Sn.data(),
                                               false, nullptr, factor, true);
    } else {
      const ftkx::lattice dom({2, 2, 2, 0}, {DW - 3, DH - 3, DD - 3, 2147483647LL}), core({2, 2, 2, t}, {DW - 3, DH - 3, DD - 3, 1}), ext({0, 0, 0}, {DW, DH, DD});
      recs = ftkx::extract_cp3dt_hip<ftkx_cp_t>(scope, t, dom, core, ext, Vc.data(), Vn.data(), Jc.data(), Jn.data(), Sc.data(), Sn.data(), factor, true);
    }
  } catch (const std::exception &e) { fprintf(stderr, "%s\n", e.what()); return 1; }
  FILE *out = fopen(argv[2], "wb");
  const uint64_t n = recs.size();
  fwrite(&n, 8, 1, out);
  if (n) fwrite(recs.data(), sizeof(ftkx_cp_t), n, out);
  fclose(out);
  return 0;
}
