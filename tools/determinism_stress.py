"""Repeat frame 0 of the C3 stream (12.8k decoded voxels: 25k lattice tiles + 24k refine tiles + 74k encoder tiles per repeat) on a
fresh map and compare every repeat with the first one, bit for bit: latents, the fold table, both cube arrays and the counters.
    python tools/determinism_stress.py [repeats]            (DIF_DECODER_PIPE=f32 for the f32-input MFMA kernels)
Found the one flaky build of round 2: with the SLP vectoriser on (v_pk_fma_f32 in decoder_fold_consts) lattice tiles come out with 16 wrong
fold constants — with the round-6 kernels 5-13 voxels per launch, every repeat (DIF_LIB=<a build without -fno-slp-vectorize>); the cause is
reproduced on its own by tools/micro/pk_fma_fold.hip (di_fusion_amd/_build.py, profiles/r06_experiments.md 5)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def run(repeats, verbose=True):
    from di_fusion_amd import synthetic as syn
    from di_fusion_amd.network import utility as U
    from di_fusion_amd.system.map import DenseIndexedMap
    dev = torch.device("cuda:0")
    scene, cfg = syn.config_c3()
    model = U.networks_from_arrays(U.load_weights_npz())
    xyz, nrm = syn.frame_points(scene, 0, syn.Intrinsic(), deg_per_frame=0.5)
    xyz, nrm = xyz.to(dev), nrm.to(dev)
    ref, bad = None, 0
    for it in range(repeats):
        m = DenseIndexedMap(model, cfg.namespace(), 29, dev, initial_capacity=32768)
        m.integrate_keyframe(xyz, nrm)
        m.extract_mesh_arrays(4, int(4e6), max_std=0.15, no_cache=False)
        B = m.last_counters["B"]
        t = m._xbuf[1]
        cur = dict(latent=m.latent_vecs[:m.n_occupied].clone(), fold=t["fold_table"][:B].clone(), sdf=t["cube_sdf"][:B].clone(),
                   std=t["cube_std"][:B].clone(), counters=dict(m.last_counters))
        if ref is None:
            ref = cur
            continue
        diffs = {k: int((cur[k] != ref[k]).sum().item()) for k in ("latent", "fold", "sdf", "std")}
        if any(diffs.values()) or cur["counters"] != ref["counters"]:
            bad += 1
            if verbose:
                vox = torch.nonzero((cur["sdf"] != ref["sdf"]).reshape(B, -1).any(1)).flatten()[:6].tolist()
                print(f"  repeat {it}: differing elements {diffs}, voxels {vox}, VH {cur['counters']['VH']} vs {ref['counters']['VH']}")
    return bad, ref["counters"]


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    if "--mc-grid-cap" in sys.argv:           # the one-pass marching cubes in ticket mode (test hook of the library)
        from di_fusion_amd import _lib
        _lib.load().dif_test_mc_grid_cap(int(sys.argv[sys.argv.index("--mc-grid-cap") + 1]))
    bad, c = run(n)
    print(f"{bad} of {n - 1} repeats differ from the first (B={c['B']} VH={c['VH']} M={c['M']})")
