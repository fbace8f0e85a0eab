"""Mask derivation tool against the oracle's restatement of the consumer rule."""
import os

import numpy as np
import pytest

from dct_pruning_amd import masks
from oracle import dct_oracle as orc


def test_compress_rate_dsl():
    assert masks.parse_compress_rate("[0.50]*7+[0.95]*5") == [0.5] * 7 + [0.95] * 5
    assert masks.parse_compress_rate("[0.10]+[0.4]*2") == [0.1, 0.4, 0.4]
    with pytest.raises(ValueError):
        masks.parse_compress_rate("[1]*3")  # the reference needs a decimal point in every rate


def test_select_index_matches_oracle_with_ties():
    rng = np.random.default_rng(0)
    imp = rng.random(192).astype(np.float32)
    imp[rng.random(192) < 0.5] = 0.0  # exact-zero ties as in the shipped imp_conv1_.npy
    for keep in (1, 96, 150, 192):
        np.testing.assert_array_equal(masks.select_index(imp, 192, keep), orc.select_index(imp, 192, keep))


def test_masks_for_dir_and_compare(tmp_path):
    rng = np.random.default_rng(1)
    a, b = tmp_path / "a", tmp_path / "b"
    a.mkdir()
    b.mkdir()
    for i, c in enumerate([64, 64, 128], 1):
        v = rng.random(c).astype(np.float32)
        np.save(a / ("imp_conv%d.npy" % i), v)
        np.save(b / ("imp_conv%d.npy" % i), v * np.float32(1 + 1e-7))  # 1-ulp-scale noise keeps the order
    np.save(a / "imp_conv10.npy", rng.random(32).astype(np.float32))
    np.save(b / "imp_conv10.npy", np.load(a / "imp_conv10.npy")[::-1].copy())
    assert masks.score_files(str(a)) == ["imp_conv1.npy", "imp_conv2.npy", "imp_conv3.npy", "imp_conv10.npy"]
    m = masks.masks_for_dir(str(a), [0.5, 0.5, 0.25, 0.5])
    assert [m[k].size for k in m] == [32, 32, 96, 16]
    assert masks.compare(m, masks.masks_for_dir(str(b), [0.5, 0.5, 0.25, 0.5])) == ["imp_conv10"]
    assert masks.main(["--imp_score", str(a), "--compress_rate", "[0.5]*4", "--compare", str(a),
                       "--out", str(tmp_path / "m.npz")]) == 0
    assert sorted(np.load(tmp_path / "m.npz").files) == sorted(m)


def test_on_the_reference_shipped_scores():
    d = "/root/reference/importance_score/googlenet_limit5"
    if not os.path.isdir(d):
        pytest.skip("reference tree not present on this box")
    m = masks.masks_for_dir(d, 0.4)
    assert len(m) == 37
    for k, v in m.items():
        imp = np.load(os.path.join(d, k + ".npy"))
        np.testing.assert_array_equal(v, orc.select_index(imp, imp.size, int(imp.size * 0.6)))
