"""Reader for the binary dumps written by oracle/_ref/ftk_ref_driver (the real reference CPU path)."""
import numpy as np

REF_REC = np.dtype([("tag", "<u8"), ("type", "<u4"), ("ordinal", "<i4"), ("timestep", "<i4"), ("_pad", "<i4"),
                    ("x", "<f8", (3,)), ("t", "<f8"), ("scalar", "<f8")])
assert REF_REC.itemsize == 64


def read_dump(path):
    raw = open(path, "rb").read()
    assert raw[:8] == b"FTKREF1\0", "not a ftk_ref_driver dump"
    off = 8
    nd, nv, D0, D1, D2, DT = np.frombuffer(raw, "<i4", 6, off); off += 24
    nf = int(np.frombuffer(raw, "<u8", 1, off)[0]); off += 8
    fs = np.frombuffer(raw, "<u8", 2 * nf, off).reshape(nf, 2); off += 16 * nf
    factors = np.zeros(DT, dtype=np.uint64)
    for step, f in fs:
        factors[int(np.int64(step))] = f
    nrec = int(np.frombuffer(raw, "<u8", 1, off)[0]); off += 8
    recs = np.frombuffer(raw, REF_REC, nrec, off).copy(); off += 64 * nrec
    n = int(nv) * int(D0) * int(D1) * int(D2)
    steps = np.frombuffer(raw, "<f8", n * int(DT), off).reshape(int(DT), n).copy()
    dims = [int(D0), int(D1)] + ([int(D2)] if nd == 3 else [])
    shape = tuple(reversed(dims)) + ((int(nv),) if nv > 1 else ())
    steps = steps.reshape((int(DT),) + shape)
    off += 8 * n * int(DT)
    curves = None
    if raw[off:off + 4] == b"CURV":
        off += 4
        nc = int(np.frombuffer(raw, "<u8", 1, off)[0]); off += 8
        curves = []
        for _ in range(nc):
            loop, npts = np.frombuffer(raw, "<i4", 2, off); off += 8
            tags = np.frombuffer(raw, "<u8", int(npts), off).copy(); off += 8 * int(npts)
            curves.append((int(loop), tags))
    pp = None
    if raw[off:off + 4] == b"PPCV":
        off += 4
        nc = int(np.frombuffer(raw, "<u8", 1, off)[0]); off += 8
        pp = []
        PT = np.dtype([("tag", "<u8"), ("type", "<u4"), ("_pad", "<u4"), ("t", "<f8")])
        for _ in range(nc):
            loop, npts = np.frombuffer(raw, "<i4", 2, off); off += 8
            pts = np.frombuffer(raw, PT, int(npts), off).copy(); off += PT.itemsize * int(npts)
            pp.append((int(loop), pts))
    return dict(nd=int(nd), nv=int(nv), dims=dims, DT=int(DT), factors=factors, records=recs, steps=steps, curves=curves, pp=pp)


def write_input(path, steps, nd, nv):
    """Writes the `file`-mode input of ftk_ref_driver."""
    steps = [np.ascontiguousarray(s, dtype="<f8") for s in steps]
    shp = steps[0].shape[:nd]
    D = [shp[nd - 1 - d] for d in range(nd)] + [1] * (3 - nd)
    with open(path, "wb") as f:
        f.write(np.array([nd, nv, D[0], D[1], D[2], len(steps)], dtype="<i4").tobytes())
        for s in steps:
            f.write(s.tobytes())
