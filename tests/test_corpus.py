"""The WHOLE shipped test set as a parity set (round 6): the 20 PhotoGen scenes x their transforms = the 90 edits of
/root/reference/test/data/photogen/photogen.json that /root/reference/test/test_diffusion_handles.py:216-225 runs.

tests/golden/g16_corpus.npz holds, per edit, what the REFERENCE's own transform_depth_pc produced on the scene's estimated depth
maps (tools/make_golden_corpus.py: SHA-256 of the correspondences, raw z-buffer mask, cleaned mask, visibility; counts; a
disparity slice).  The CPU test holds the oracle against it on one edit per scene; the GPU test runs every scene's edits as ONE
batched `reproject_edits` call through the C ABI and compares all 90: integer maps bit-exact, disparity <= 2e-3 / 255 (f64 CG vs
the reference's sparse direct solve).  This is where a z-tie, an off-frame foreground (car/edit_000: N_corr 0), or a hole of more
than 65 536 unknowns would show up (mask areas 4 357 - 50 618 px, angles -35 ... 91 degrees)."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "golden", "photogen")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def same(h, stored):
    return h == str(stored)


@pytest.fixture(scope="module")
def corpus():
    g = np.load(os.path.join(HERE, "golden", "g16_corpus.npz"))
    with open(os.path.join(ROOT, "photogen.json")) as f:
        test_set = json.load(f)
    return g, test_set


def _tf(t):
    from diffusionhandles_amd import scene_io as S
    kw = S.transform_args(t)
    return kw["rot_angle"], kw["rot_axis"], kw["translation"]


def test_corpus_fixture_is_complete_and_inputs_unchanged(corpus):
    from diffusionhandles_amd import scene_io as S
    g, test_set = corpus
    assert len(test_set) == 20 and sum(len(v) for v in test_set.values()) == 90
    for scene, edits in test_set.items():
        sc = S.load_scene_geometry(os.path.join(ROOT, scene), 512)
        assert same(sha(sc["depth"].numpy()), g[f"{scene}/depth_sha"]), scene
        assert same(sha(sc["bg_depth"].numpy()), g[f"{scene}/bg_depth_sha"]), scene
        assert same(sha(np.packbits(sc["fg_mask"].numpy() != 0)), g[f"{scene}/mask_sha"]), scene
        for e in edits:
            assert e in sc["transforms"] and f"{scene}/{e}/corr_sha" in g.files, (scene, e)


def test_oracle_matches_reference_on_one_edit_per_scene(corpus):
    """The oracle (NumPy) against the reference's stored results: the LAST listed edit of every scene (the largest motions)."""
    from diffusionhandles_amd import scene_io as S
    from oracle import depth_ref as D
    g, test_set = corpus
    for scene, edits in test_set.items():
        sc = S.load_scene_geometry(os.path.join(ROOT, scene), 512)
        name = edits[-1]
        ang, axis, tr = _tf(sc["transforms"][name])
        disp, corr, dbg = D.transform_depth_pc(sc["depth"], sc["bg_depth"], sc["fg_mask"], D.intrinsics_f32(), rot_angle=ang,
                                               rot_axis=[float(v) for v in axis], translation=[float(v) for v in tr], return_debug=True)
        key = f"{scene}/{name}"
        assert corr.shape[0] == int(g[key + "/n_corr"]), key
        assert same(sha(corr.numpy().astype(np.int64)), g[key + "/corr_sha"]), key
        assert same(sha(np.packbits(dbg["raw_mask"] != 0)), g[key + "/raw_mask_sha"]), key
        assert same(sha(np.packbits(dbg["cleaned"] != 0)), g[key + "/cleaned_sha"]), key
        assert np.allclose(disp[0, 0].numpy()[::17, ::19], g[key + "/disp_slice"], atol=1e-4, rtol=0), key


@pytest.mark.gpu
def test_all_90_shipped_edits_bit_exact_vs_reference_golden(corpus):
    from diffusionhandles_amd import depth_transform as DT
    from diffusionhandles_amd import scene_io as S
    from oracle import depth_ref as D
    g, test_set = corpus
    assert torch.cuda.is_available()
    dev = torch.device("cuda:0")
    n_edits, worst_disp, misses = 0, 0.0, []
    for scene, edits in test_set.items():
        sc = S.load_scene_geometry(os.path.join(ROOT, scene), 512)
        tf = [_tf(sc["transforms"][e]) for e in edits]
        out, dbg = DT.reproject_edits(sc["depth"].to(dev), sc["bg_depth"].to(dev), sc["fg_mask"].to(dev), D.intrinsics_f32(), tf,
                                      return_debug=True)
        for i, name in enumerate(edits):
            key = f"{scene}/{name}"
            disp, corr = out[i]
            n_edits += 1
            bad = []
            if corr.shape[0] != int(g[key + "/n_corr"]) or not same(sha(corr.numpy().astype(np.int64)), g[key + "/corr_sha"]):
                bad.append(f"correspondences (N {corr.shape[0]} vs {int(g[key + '/n_corr'])})")
            if not same(sha(np.packbits(dbg["raw_mask"][i].cpu().numpy() != 0)), g[key + "/raw_mask_sha"]):
                bad.append("raw mask")
            if not same(sha(np.packbits(dbg["clean_mask"][i].cpu().numpy() != 0)), g[key + "/cleaned_sha"]):
                bad.append("cleaned mask")
            if not same(sha(np.packbits(dbg["vis"][i].cpu().numpy() != 0)), g[key + "/vis_sha"]):
                bad.append("visibility")
            d = disp[0, 0].cpu().numpy()
            de = float(np.abs(d[::17, ::19] - g[key + "/disp_slice"]).max())
            worst_disp = max(worst_disp, de)
            if de > 2e-3:
                bad.append(f"disparity slice {de:.2e}")
            if abs(float(d.astype(np.float64).sum()) - float(g[key + "/disp_sum"])) > 1.0 + 1e-6 * abs(float(g[key + "/disp_sum"])):
                bad.append("disparity sum")
            if bad:
                misses.append(f"{key}: {', '.join(bad)}")
    print(f"shipped corpus: {n_edits} edits of {len(test_set)} scenes, {n_edits - len(misses)} bit-exact on every integer map, worst disparity "
          f"slice difference {worst_disp:.2e} (gate 2e-3 on a [0, 255] scale)")
    assert n_edits == 90
    assert not misses, misses
