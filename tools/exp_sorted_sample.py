import math, sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import pixell_jl_amd as pj
dev = torch.device("cuda:0")
shape, wcs = pj.fullsky_geometry(2 * math.pi / 43200)
m = pj.Enmap(torch.empty((shape[1], shape[0]), dtype=torch.float64, device=dev), wcs)
pj.fill_random_(m.data, 1234)
n = 200_000_000
sky = torch.empty((n, 2), dtype=torch.float64, device=dev)
pj.fill_sphere_points_(sky, 42)
def t(f, reps=5):
    f(); torch.cuda.synchronize(); ts=[]
    for _ in range(reps):
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts)//2]
print("random order       : %.2f ms  (%.1f Gpts/s)" % (t(lambda: pj.sample_bilinear(m, sky)), n / t(lambda: pj.sample_bilinear(m, sky)) / 1e6))
# sort by DEC only (coarse: 64-row bands)
pix = pj.sky2pix(m, sky)
band = (pix[:, 1] / 64).to(torch.int32)
order = torch.argsort(band)
sky_b = sky[order].contiguous()
ms = t(lambda: pj.sample_bilinear(m, sky_b)); print("sorted by 64-row band: %.2f ms  (%.1f Gpts/s)" % (ms, n / ms / 1e6))
# sort by (band of 64 rows, band of 512 cols)
key = band.to(torch.int64) * 128 + (pix[:, 0] / 512).to(torch.int64)
order = torch.argsort(key)
sky_t = sky[order].contiguous()
ms = t(lambda: pj.sample_bilinear(m, sky_t)); print("sorted by 64x512 tile : %.2f ms  (%.1f Gpts/s)" % (ms, n / ms / 1e6))
torch.cuda.synchronize(); t0=time.time(); order = torch.argsort(key); torch.cuda.synchronize(); print("torch argsort of keys: %.1f ms" % ((time.time()-t0)*1e3))
