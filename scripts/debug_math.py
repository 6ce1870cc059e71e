import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from snowmocap_amd import _lib
ctx = _lib.scratch_context()
rng = np.random.default_rng(0)
for lo, hi in ((-300, 300), (-30, 30), (-3, 3), (0, 1)):
    n = 200000
    x = rng.uniform(1, 10, n) * 10.0 ** rng.integers(lo, hi + 1, n)
    xs = x * rng.choice([-1.0, 1.0], n)
    r2 = np.empty(n); r1 = np.empty(n); q1 = np.empty(n)
    _lib.check(_lib.lib().snowtri_fastmath_probe(ctx.handle, n, _lib.ptr(xs), _lib.ptr(r2), _lib.ptr(r1), _lib.ptr(q1)), "probe")
    e2 = np.abs(r2 * xs - 1); e1 = np.abs(r1 * xs - 1)
    _lib.check(_lib.lib().snowtri_fastmath_probe(ctx.handle, n, _lib.ptr(x), _lib.ptr(r2), _lib.ptr(r1), _lib.ptr(q1)), "probe")
    eq = np.abs(q1 * q1 * x - 1)
    print(f"decades [{lo},{hi}]: rcp_nr2 max rel {np.nanmax(e2):.2e} (nan {np.isnan(e2).sum()}), rcp_nr1 {np.nanmax(e1):.2e} (nan {np.isnan(e1).sum()}), rsq_nr1 {np.nanmax(eq)/2:.2e} (nan {np.isnan(eq).sum()})")
    bad = np.argsort(-np.nan_to_num(e2, nan=1e9))[:3]
    print("   worst rcp_nr2 inputs:", xs[bad], e2[bad])
