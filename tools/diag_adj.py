import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import cases
from adjointnonlinearraytracing_amd import drrt
gpu = torch.device("cuda:0")
drrt.options.check_failed = False
T = drrt.TracerC()
shape, n = (65, 65, 65), 6000
D_, H_, W_ = shape
h = 1.0 / (max(shape) - 1); ds = h / 2
rng = np.random.default_rng(21)
rif = torch.from_numpy((1.0 + 0.3 * rng.random(shape, dtype=np.float32)).astype(np.float32)).to(gpu)
res = (W_, H_, D_)
ext = np.array([(W_ - 1) * h, (H_ - 1) * h, (D_ - 1) * h], np.float32)
pos = (rng.uniform(0.02, 0.98, (n, 3)) * ext).astype(np.float32); pos[:, 1] = 0.0
vel = rng.normal(0, 0.3, (n, 3)).astype(np.float32); vel[:, 1] = 1.0
vel /= np.linalg.norm(vel, axis=1, keepdims=True)
dx = rng.normal(size=(n, 3)).astype(np.float32); dv = rng.normal(size=(n, 3)).astype(np.float32)
P, V_, DX, DV = (torch.from_numpy(a).to(gpu) for a in (pos, vel, dx, dv))
xt, vt = T.trace(rif, res, P, V_, h, ds)
order = drrt.last_order
out = {}
for name, kw in (("flat", {}), ("flat2", {}), ("direct", {"direct_atomics": True}), ("legacy", {"legacy_adjoint": True}), ("nofit", {"_exp": 7})):
    for k, v in kw.items():
        if k != '_exp': setattr(drrt.options, k, v)
    drrt._EXPERIMENT = kw.get('_exp', 0)
    out[name] = T.backtrace(rif, res, xt, vt, DX, DV, h, ds, order=order).cpu().numpy().astype(np.float64)
    for k, v in kw.items():
        if k != '_exp': setattr(drrt.options, k, False)
for a in out:
    for b in out:
        if a < b: print(a, b, "rel_l2", cases.rel_l2(out[a], out[b]))
d = out["flat"] - out["direct"]
i = np.argsort(-np.abs(d).ravel())[:8]
print("largest diffs flat-direct:", [(np.unravel_index(k, d.shape), float(d.ravel()[k]), float(out["direct"].ravel()[k])) for k in i])
print("sum flat", out["flat"].sum(), "sum direct", out["direct"].sum(), "sum legacy", out["legacy"].sum())
