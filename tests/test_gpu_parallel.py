"""GPU: record export / merge kernels against the oracle (single process; the collective itself is covered by gloo on CPU)."""
import numpy as np
import pytest
import torch

from di_fusion_amd import synthetic as syn
from tests.conftest import GOLDEN

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def test_export_merge_records_vs_oracle(gpu_model, oracle_net):
    from di_fusion_amd.system.map import DenseIndexedMap
    from oracle import difusion_oracle as O
    cfg = syn.MapConfig((-1.6,) * 3, (1.6,) * 3, 0.4)
    intr = syn.Intrinsic().scaled(0.125)
    scene = syn.Scene(kind="sphere", radius=1.3)
    gm, om = [], []
    for rank in range(2):
        m = DenseIndexedMap(gpu_model, cfg.namespace(), 29, DEV, initial_capacity=1024)
        o = O.OracleMap(oracle_net, cfg.bound_min, cfg.bound_max, cfg.voxel_size)
        for f in range(2):
            xyz, nrm = syn.frame_points(scene, f, intr, deg_per_frame=20.0, phase_deg=90.0 * rank)
            m.integrate_keyframe(xyz.to(DEV), nrm.to(DEV))
            o.integrate_keyframe(xyz.numpy(), nrm.numpy())
        gm.append(m); om.append(o)
    recs = [m.export_records() for m in gm]
    for r, o in zip(recs, om):
        want = O.export_records(o)
        got = r.cpu().numpy()
        assert np.array_equal(got[:, :3], want[:, :3])                     # lin ids and weights bit-exact
        assert np.abs(got[:, 3:].view(np.float32) - want[:, 3:].view(np.float32)).max() < 1e-3
    g = DenseIndexedMap(gpu_model, cfg.namespace(), 29, DEV, initial_capacity=1024)
    go = O.OracleMap(oracle_net, cfg.bound_min, cfg.bound_max, cfg.voxel_size)
    for r in recs:
        g.merge_records(r)
        O.merge_records(go, r.cpu().numpy())
    n = go.n_occupied
    assert g.n_occupied == n
    assert np.array_equal(g.latent_vecs_pos[:n].cpu().numpy(), go.latent_vecs_pos[:n])
    assert np.array_equal(g.voxel_obs_count[:n].cpu().numpy(), go.voxel_obs_count[:n])
    assert np.abs(g.latent_vecs[:n].cpu().numpy() - go.latent_vecs[:n]).max() < 1e-5
    assert np.array_equal(g.updated_vec_id.cpu().numpy(), go.updated_vec_id)
    # the merged map meshes
    v, vid, vs = g.extract_mesh_arrays(4, int(4e6), max_std=0.15)
    assert v.shape[0] > 0
