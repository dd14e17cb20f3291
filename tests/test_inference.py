"""Sliding-window inference: invariants of the tiling restated from nnU-Net v2.0 and the
reference's own inference behaviour (single tensor without deep supervision, NoMirroring trainers)."""
import itertools

import numpy as np
import pytest
import torch
from torch import nn

import model_cases as mc
from nextou_amd import inference as inf


def test_steps_cover_the_image_and_respect_the_step():
    for size, tile, step in (((96, 80), (64, 64), 0.5), ((64, 64), (64, 64), 0.5), ((130, 300, 17), (32, 128, 17), 0.5),
                             ((65,), (64,), 1.0), ((200,), (64,), 0.25)):
        steps = inf.compute_steps_for_sliding_window(size, tile, step)
        for s, n, t in zip(steps, size, tile):
            assert s[0] == 0 and s[-1] == n - t and s == sorted(set(s))
            assert all(b - a <= int(np.ceil(t * step)) for a, b in zip(s, s[1:]))
    assert inf.compute_steps_for_sliding_window((96, 80), (64, 64), 0.5) == [[0, 32], [0, 16]]
    with pytest.raises(AssertionError):
        inf.compute_steps_for_sliding_window((10,), (64,), 0.5)


def test_gaussian_matches_scipy_nd_filter():
    from scipy.ndimage import gaussian_filter
    for tile in ((16, 24), (8, 12, 10)):
        tmp = np.zeros(tile)
        tmp[tuple(i // 2 for i in tile)] = 1
        want = gaussian_filter(tmp, [i / 8 for i in tile], 0, mode="constant", cval=0)
        want = want / want.max()
        want[want == 0] = want[want != 0].min()
        got = inf.compute_gaussian(tile).double().numpy()
        np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-12)
        assert got.max() == 1.0 and got.min() > 0


def test_mirror_subsets():
    assert inf.mirror_axis_subsets(None) == [()]
    assert inf.mirror_axis_subsets((0, 1)) == [(), (0,), (1,), (0, 1)]
    assert len(inf.mirror_axis_subsets((0, 1, 2))) == 8


class _Pointwise(nn.Module):
    """Stand-in network: per-voxel affine map, optionally position-dependent inside the patch."""

    def __init__(self, patch, position_dependent):
        super().__init__()
        self.decoder = nn.Module()
        self.decoder.deep_supervision = True
        g = torch.Generator().manual_seed(0)
        self.ramp = torch.rand(patch, generator=g) if position_dependent else torch.zeros(patch)
        self.calls = []

    def forward(self, x):
        assert not self.training and self.decoder.deep_supervision is False
        self.calls.append(x.shape[0])
        return torch.cat([2 * x + self.ramp, -x + 1], 1)


def _naive(net_fn, image, patch, step, gaussian, mirror_axes):
    """Independent restatement with plain loops, one forward per tile and mirror copy."""
    spatial = image.shape[1:]
    steps = inf.compute_steps_for_sliding_window(spatial, patch, step)
    w = inf.compute_gaussian(patch) if gaussian else torch.ones(patch)
    out = cnt = None
    for origin in itertools.product(*steps):
        sl = tuple(slice(o, o + p) for o, p in zip(origin, patch))
        tile = image[(slice(None),) + sl][None]
        pred = net_fn(tile)
        flips = inf.mirror_axis_subsets(mirror_axes)[1:]
        for f in flips:
            dims = [a + 2 for a in f]
            pred = pred + torch.flip(net_fn(torch.flip(tile, dims)), dims)
        pred = pred[0] / (len(flips) + 1)
        if out is None:
            out = torch.zeros((pred.shape[0],) + tuple(spatial))
            cnt = torch.zeros(tuple(spatial))
        out[(slice(None),) + sl] += pred * w
        cnt[sl] += w
    return out / cnt


@pytest.mark.parametrize("mirror", [None, (0, 1)])
def test_sliding_window_equals_naive_loops(mirror):
    patch = (16, 24)
    net = _Pointwise(patch, position_dependent=True).train()
    image = torch.randn(1, 40, 50)
    got = inf.predict_sliding_window(net, image, patch, 0.5, True, mirror, batch_size=5)
    assert net.training and net.decoder.deep_supervision is True          # state restored
    assert max(net.calls) == 5                                            # tiles and mirror copies were batched
    net.eval()
    net.decoder.deep_supervision = False
    want = _naive(net, image, patch, 0.5, True, mirror)
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-6)


def test_position_independent_network_is_reproduced_exactly_and_padding_is_cropped():
    patch = (8, 8, 8)
    net = _Pointwise(patch, position_dependent=False)
    image = torch.randn(1, 20, 9, 13)
    got = inf.predict_sliding_window(net, image, patch, 0.5, True, (0, 1, 2), batch_size=16)
    torch.testing.assert_close(got, torch.cat([2 * image, -image + 1], 0), rtol=1e-5, atol=1e-6)
    small = torch.randn(1, 5, 8, 6)                                        # smaller than the patch: zero-padded, cropped
    got = inf.predict_sliding_window(net, small, patch, 0.5, True, None)
    assert got.shape == (2, 5, 8, 6)
    torch.testing.assert_close(got, torch.cat([2 * small, -small + 1], 0), rtol=1e-5, atol=1e-6)


def test_nextou_single_patch_inference_equals_eval_forward(cpu_checker):
    """image == patch, no mirroring: the sliding window is one eval-mode forward with deep supervision off
    (reference NexToU_Encoder_Decoder.py:333-337), and the NoMirroring trainer's policy reaches the predictor."""
    from nextou_amd.harness import StandaloneTrainerBase, config_2d_nextou
    from nextou_amd.nnUNetTrainer.nnUNetTrainer_NexToU import nnUNetTrainer_NexToU
    from nextou_amd.nnUNetTrainer.nnUNetTrainer_NexToU_NoMirroring import nnUNetTrainer_NexToU_NoMirroring
    torch.manual_seed(0)
    net = mc.build_model(mc.TINY_2D)
    image = torch.randn(1, 64, 64)
    got = inf.predict_sliding_window(net, image, (64, 64), 0.5, True, None)
    net.eval()
    net.decoder.deep_supervision = False
    with torch.no_grad():
        want = net(image[None])[0]
    assert torch.equal(got, want)
    # mirroring TTA of the real model = mean over the four flipped forwards
    tta = inf.predict_sliding_window(net, image, (64, 64), 0.5, True, (0, 1), batch_size=4)
    with torch.no_grad():
        ref = sum(torch.flip(net(torch.flip(image[None], [d + 2 for d in f])), [d + 2 for d in f])
                  for f in inf.mirror_axis_subsets((0, 1)))[0] / 4
    torch.testing.assert_close(tta, ref, rtol=1e-4, atol=1e-5)

    cfg = config_2d_nextou(patch_size=(64, 64), base=8, max_features=64, n_stages=5)
    for cls, axes in ((nnUNetTrainer_NexToU, (0, 1)), (nnUNetTrainer_NexToU_NoMirroring, None)):
        if not issubclass(cls, StandaloneTrainerBase):
            pytest.skip("nnunetv2 is installed: the real trainer base drives inference")
        t = cls(cfg, 3, log=None)
        t.configure_rotation_dummyDA_mirroring_and_inital_patch_size()
        assert t.inference_allowed_mirroring_axes == axes
