"""On-disk frame format round trip (README.md:55-62, app/data_engine.py:53-158) on synthetic scenes."""
import json

import numpy as np


def test_pickle_round_trip_and_ee_relabel(tmp_path):
    import mrcc_amd
    from mrcc_amd.app.data_engine import PickleDataEngine, get_ee_idx, write_frame_pickle

    items = []
    scenes = []
    for i, seed in enumerate((3, 1, 2)):
        sc = mrcc_amd.synth.gen_scene(seed, n_bg=2000, n_arm=300, n_ee=600)
        scenes.append(sc)
        path = tmp_path / f"{10 - i}.pickle"
        write_frame_pickle(path, sc)
        items.append({"filepath": path.name, "position": sc["position"]})
    split = tmp_path / "split.json"
    split.write_text(json.dumps({"test": items}))
    eng = PickleDataEngine(str(split), split="test", cyclic=False)
    assert len(eng) == 3
    raws = [eng.get_raw() for _ in range(3)]
    assert eng.get_raw() is None
    keys = [(r.other["position"], int(r.other["filepath"].split("/")[-1].split(".")[0])) for r in raws]
    assert keys == sorted(keys)  # (position, numeric file name) order
    for r in raws:
        sc = next(s for s in scenes if np.array_equal(s["points"], r.points))
        assert np.allclose(r.pose, sc["pose"], atol=1e-6)  # stored xyzw, returned wxyz
        assert np.allclose(r.ee2base_pose, sc["ee2base_pose"], atol=1e-6)
        # the EE box test re-derives label 2 from the arm-labelled points around the pose
        ee_true = sc["segmentation"] == 2
        ee_got = r.segmentation == 2
        assert (ee_true & ee_got).sum() / ee_true.sum() > 0.95
        assert ((r.segmentation == 0) == (sc["segmentation"] == 0)).all()
    eng2 = PickleDataEngine(str(split), split="test", cyclic=True)
    d = eng2.get()
    assert d.points.shape[1] == 3 and d.gt_pose.shape == (7,) and abs(np.linalg.norm(d.gt_pose[3:]) - 1) < 1e-5
    idx = get_ee_idx(scenes[0]["points"], scenes[0]["pose"], switch_w=False)
    assert len(idx) > 0
