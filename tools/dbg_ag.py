import sys
sys.path.insert(0,'/root/repo')
import numpy as np
from kimera_semantics_amd import binding as B, synth
from oracle import oracle_py as O
from tests.util import COMMON, NO_EARLY_OUT, compare_maps
kw = dict(COMMON, method=1, max_consecutive_ray_collisions=NO_EARLY_OUT, enable_anti_grazing=1)
sc = synth.make_scene("room")
for pipe in (0,1):
    o = O.Oracle(O.default_config(**kw))
    h = B.HipIntegrator(B.default_config(max_tiles=4096, max_points=1 << 15, pipeline_frames=pipe, **kw))
    for k in range(5):
        f = synth.render_frame(sc, synth.trajectory_pose(3 * k), 160, 120, seed=900 + k)
        so=o.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
        sh=h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
        if pipe: sh=h.flush()
        rep=compare_maps(o,h,exact=False)
        print('pipe',pipe,'frame',k,'updates',so.n_voxel_updates,sh.n_voxel_updates,'rays',so.n_rays_cast,sh.n_rays_cast,'label mism',rep['label_mismatches'],'touched',rep['oracle_touched'],rep['hip_touched'])
