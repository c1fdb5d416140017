import sys, time, numpy as np
sys.path.insert(0, '.')
from kimera_semantics_amd import binding as B, synth
from oracle import oracle_py as O
from tests.util import COMMON, NO_EARLY_OUT, compare_maps, small_frame
f = small_frame(seed=1)
for method, mc in ((1, 2), (0, NO_EARLY_OUT), (0, 2)):
    kw = dict(COMMON, method=method, max_consecutive_ray_collisions=mc)
    o = O.Oracle(O.default_config(**kw)); h = B.HipIntegrator(B.default_config(max_tiles=4096, **kw))
    so = o.integrate(f.T_G_C, f.xyz, f.rgba, f.labels); sh = h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
    print("method", method, "mc", mc, "oracle", so.n_valid_points, so.n_rays_cast, so.n_voxel_updates, "hip", sh.n_valid_points, sh.n_rays_cast, sh.n_voxel_updates)
    try:
        print(compare_maps(o, h, exact=(mc != 2 or method == 1)))
    except AssertionError as e:
        print("MISMATCH", str(e)[:2000])
        print(compare_maps(o, h, exact=False))
