#!/usr/bin/env python3
"""Wall-clock rate of the C++ drop-in adapter (kimera::HipSemanticTsdfIntegrator behind the
TsdfIntegratorBase virtual) on 640x480 host clouds: strict per-frame layer sync (kEveryFrame)
vs on-demand sync with pipelined frames.  Needs a GPU; numbers go to INTEGRATION.md."""
import os
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kimera_semantics_amd import synth  # noqa: E402

DEMO = os.path.join(ROOT, "kimera_semantics_amd", "host", "adapter_demo")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
sc = synth.make_scene("room")
tmp = tempfile.mkdtemp(prefix="ks_adapter_")
fin, fout, csv = (os.path.join(tmp, x) for x in ("in.bin", "out.bin", "labels.csv"))
with open(csv, "w") as fh:   # the reference's label CSV format (name,red,green,blue,alpha,id)
    fh.write("name,red,green,blue,alpha,id\n")
    for i, (r, g, b, a) in enumerate(synth.default_label_colors()[:21]):
        fh.write(f"label{i},{int(r)},{int(g)},{int(b)},{int(a)},{i}\n")
with open(fin, "wb") as fh:
    fh.write(struct.pack("<I", n))
    for k in range(n):
        f = synth.render_frame(sc, synth.trajectory_pose(10 + k), 640, 480, seed=10 + k)
        fh.write(f.T_G_C.astype("<f4").tobytes())
        fh.write(struct.pack("<I", len(f.xyz)))
        fh.write(f.xyz.astype("<f4").tobytes())
        fh.write(f.rgba.tobytes())
for method in ("fast", "merged"):
    for pipe in ("0", "1"):
        res = subprocess.run([DEMO, method, csv, fin, fout, "1", "2", "-1", pipe], capture_output=True, text=True)
        print(method, "pipeline" if pipe == "1" else "strict  ", "|", " | ".join(l for l in res.stdout.splitlines()), res.stderr[-600:])
