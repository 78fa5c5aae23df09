for c in c3 c4 c2; do python tools/tracker_api_bench.py $c 2>/dev/null | tail -1; done
python - <<'PY'
# the same series through one tracker over 2 and 4 contexts of the one GPU (threads + queues: overhead check, no speed-up expected on one device)
import sys, time
sys.path.insert(0, ".")
import torch, ftk_amd
from ftk_amd import synthetic, tslab
dims, nt = (256, 256, 256), 16
dev = torch.device("cuda", 0)
slices = [synthetic.generate("moving_extremum_3d", dims, t, nt, torch, dev) for t in range(nt)]
torch.cuda.synchronize()
for ids, block in (([0, 0], 2), ([0, 0, 0, 0], 2)):
    for rep in range(2):
        tr = ftk_amd.CriticalPointTracker3DRegular(device_ids=ids, block=block)
        tr.set_scalar_field_source(ftk_amd.SOURCE_GIVEN); tr.set_vector_field_source(ftk_amd.SOURCE_DERIVED)
        tr.set_jacobian_field_source(ftk_amd.SOURCE_DERIVED); tr.set_jacobian_symmetric(True)
        tr.set_domain([2] * 3, [d - 3 for d in dims]); tr.set_array_domain([0] * 3, list(dims))
        tr.set_tag_mode(ftk_amd.TAG_EXACT64); tr.initialize()
        t0 = time.perf_counter()
        for k in range(nt):
            tr.push_scalar_field_snapshot(slices[k])
            if k != 0: tr.advance_timestep()
            if k == nt - 1: tr.update_timestep()
        tr.sync()
        dt = time.perf_counter() - t0
        recs, o, ts = tr.get_critical_points(); tr.close()
    print("c3 one tracker over", len(ids), "contexts (same GPU), block", block, ": %.3f ms per timestep, %d records" % (dt * 1e3 / nt, len(recs)))
PY
