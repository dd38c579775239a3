"""Where the host time of one `sdf_hg` call goes (tools/bench_query.py's setting)."""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from di_fusion_amd import synthetic as syn
from di_fusion_amd.network import utility as net_util
from di_fusion_amd.stream import FusionStream
from di_fusion_amd.system import tracker as T

dev = torch.device("cuda:0")
scene, cfg = syn.config_c3()
model = net_util.networks_from_arrays(net_util.load_weights_npz())
st = FusionStream(model, scene, cfg, syn.Intrinsic(), dev, 30, deg_per_frame=0.5)
for i in range(30):
    st.step(i, "none")
obs, R, t = syn.frame_cloud_camera(scene, 15, syn.Intrinsic().scaled(0.5), device=dev)
last, delta = T.Pose(*syn.orbit_pose(14)), T.Pose()
for _ in range(5):
    T.sdf_hg(st.map, obs, last, delta, "huber", 5.0)
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    T.sdf_hg(st.map, obs, last, delta, "huber", 5.0)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
