"""Scores and prune masks of the GPU path against the CPU oracle on IDENTICAL activations (the
forward pass runs once; every hooked tensor is scored by both), for all seven nets at a reduced
batch. The full configuration (ResNet-50 / 224x224 / batch 256 / limit 5) is the same code run as a
tool: `python tests/mask_parity.py`, summary under profiles/. Bars: scores within 1e-4 relative
(BASELINE.json), dead channels exactly +0.0, masks of the consumer rule (utils/load_models.py:40-41)
strictly equal - no near-tie allowance: the activations are the same bytes on both sides."""
import pytest

import mask_parity

pytestmark = pytest.mark.gpu

CASES = [("vgg_16_bn", 32, 5, None), ("resnet_56", 8, 2, None), ("resnet_110", 4, 1, None), ("densenet_40", 8, 2, None),
         ("googlenet", 8, 2, None), ("resnet_50", 4, 2, None), ("u2netp", 2, 1, 288), ("u2netp", 1, 1, 320)]


@pytest.mark.parametrize("net,bs,limit,size", CASES)
def test_scores_and_masks_on_identical_activations(net, bs, limit, size):
    summary, files = mask_parity.run(net, bs, limit, input_size=size)
    bad = [f for f in files if not all(f["masks_equal"].values())]
    assert not bad, bad[:3]
    assert summary["max_rel"] <= 1e-4, summary
    assert summary["all_dead_plus_zero"], summary
    assert summary["files"] == sum(len(p.files) for p in mask_parity.schedules.SCHEDULES[net]())
    if net in mask_parity.README_RATES:
        assert summary["readme_masks_checked"] > 0 and summary["readme_masks_equal"]
