import sys
sys.path.insert(0, '.')
import torch
from di_fusion_amd import synthetic as syn
from di_fusion_amd.network import utility as net_util
from di_fusion_amd.stream import FusionStream
dev = torch.device('cuda:0')
scene, cfg = syn.config_c3()
model = net_util.networks_from_arrays(net_util.load_weights_npz())
st = FusionStream(model, scene, cfg, syn.Intrinsic(), dev, 80, deg_per_frame=0.5)
for i in range(80):
    st.step(i, 'none')
torch.cuda.synchronize()
