"""Scene (LBVH) build timing at a given tile size."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import horayzon_amd as hz
from horayzon_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3601
g = synth.fractal_tile(n=n, offset=16)
import torch
v = torch.from_numpy(g["vert_grid"]).to("cuda:0")
for rep in range(3):
    t = time.time()
    sc = hz.Scene.create(v, n, n)
    print(n, "wall %.3f" % (time.time() - t), json.dumps({k: sc.stats[k] for k in ("t_bvh_s", "t_h2d_s", "bvh_height", "scene_bytes")}), flush=True)
    sc.close()
