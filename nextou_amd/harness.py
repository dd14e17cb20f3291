"""Standalone stand-ins for the parts of nnU-Net v2 that call the NexToU plug-in surface.

nnU-Net v2.0 (CLI, dataloading, augmentation, planning, checkpointing, logging …) is external to
the reference and out of scope.  This module reproduces only the *calls* a trainer
makes on the plug-ins — build the network from plans, build the loss, run forward / loss / backward
/ SGD — so that the path can be driven, tested and benchmarked without an nnU-Net installation.
With nnU-Net installed the trainer classes in ``nextou_amd.nnUNetTrainer`` derive from the real
``nnUNetTrainer`` instead and nothing here is used.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F


@dataclass
class SimpleConfigurationManager:
    """The plans fields ``nnUNetTrainer_NexToU.build_network_architecture`` reads
    (reference nnUNetTrainer_NexToU.py:24-27,76-82)."""
    patch_size: Sequence[int]
    conv_kernel_sizes: Sequence[Sequence[int]]
    pool_op_kernel_sizes: Sequence[Sequence[int]]
    UNet_base_num_features: int = 32
    unet_max_num_features: int = 320
    n_conv_per_stage_encoder: Sequence[int] = ()
    n_conv_per_stage_decoder: Sequence[int] = ()
    batch_size: int = 2
    batch_dice: bool = False

    def __post_init__(self):
        n = len(self.conv_kernel_sizes)
        if not self.n_conv_per_stage_encoder:
            self.n_conv_per_stage_encoder = [2] * n
        if not self.n_conv_per_stage_decoder:
            self.n_conv_per_stage_decoder = [2] * (n - 1)


@dataclass
class SimpleLabelManager:
    num_segmentation_heads: int
    ignore_label: Optional[int] = None
    has_regions: bool = False


@dataclass
class SimplePlansManager:
    label_manager: SimpleLabelManager

    def get_label_manager(self, dataset_json):
        return self.label_manager


# topology of the reference's example plans (nnUNetPlans.json:338-401 "3d_fullres", :26-154 "2d")
KERNELS_3D = [[1, 3, 3]] + [[3, 3, 3]] * 5
STRIDES_3D = [[1, 1, 1], [1, 2, 2]] + [[2, 2, 2]] * 4


def config_3d_fullres_nextou(patch_size=(64, 224, 192), base=33, max_features=324, batch_size=2):
    """BASELINE.json configs[1] / nnUNetPlans.json:426-435 ('3d_fullres_nextou')."""
    return SimpleConfigurationManager(patch_size=list(patch_size), conv_kernel_sizes=KERNELS_3D,
                                      pool_op_kernel_sizes=STRIDES_3D, UNet_base_num_features=base,
                                      unet_max_num_features=max_features, batch_size=batch_size)


def config_2d_nextou(patch_size=(512, 512), base=32, max_features=512, n_stages=7, batch_size=1):
    """BASELINE.json configs[0] (2-D, 7 stages, SURVEY.md §8d cfg 1)."""
    return SimpleConfigurationManager(patch_size=list(patch_size), conv_kernel_sizes=[[3, 3]] * n_stages,
                                      pool_op_kernel_sizes=[[1, 1]] + [[2, 2]] * (n_stages - 1),
                                      UNet_base_num_features=base, unet_max_num_features=max_features,
                                      batch_size=batch_size)


class StandaloneTrainerBase:
    """The slice of ``nnunetv2.training.nnUNetTrainer.nnUNetTrainer`` the plug-ins touch
    (SURVEY.md §8b 'trainer API')."""

    initial_lr, weight_decay, momentum = 1e-2, 3e-5, 0.99

    def __init__(self, configuration_manager: SimpleConfigurationManager, num_classes: int,
                 num_input_channels: int = 1, device: torch.device = torch.device("cpu"),
                 ignore_label: Optional[int] = None, is_ddp: bool = False, enable_deep_supervision: bool = True,
                 log=print):
        self.configuration_manager = configuration_manager
        self.label_manager = SimpleLabelManager(num_classes, ignore_label)
        self.plans_manager = SimplePlansManager(self.label_manager)
        self.dataset_json = {}
        self.num_input_channels = num_input_channels
        self.device = device
        self.is_ddp = is_ddp
        self.enable_deep_supervision = enable_deep_supervision
        self._log = log
        self.network = self.loss = self.optimizer = None
        # nnU-Net's default: mirror along every spatial axis at test time; the *_NoMirroring plug-ins set None
        self.inference_allowed_mirroring_axes = tuple(range(len(configuration_manager.patch_size)))

    # -- hooks the plug-ins use -----------------------------------------------------------------
    def print_to_log_file(self, *args, **kwargs):
        if self._log is not None:
            self._log(*args)

    def _get_deep_supervision_scales(self):
        pools = np.vstack(self.configuration_manager.pool_op_kernel_sizes)
        return list(list(i) for i in 1 / np.cumprod(pools, axis=0))[:-1]

    def _build_loss(self):
        """nnU-Net's default: Dice + CE under the deep-supervision wrapper."""
        from .loss.nnunet_losses import DeepSupervisionWrapper, MemoryEfficientSoftDiceLoss, \
            RobustCrossEntropyLoss, softmax_helper_dim1
        from torch import nn

        class _DCandCE(nn.Module):
            def __init__(self, batch_dice, ddp):
                super().__init__()
                self.ce = RobustCrossEntropyLoss()
                self.dc = MemoryEfficientSoftDiceLoss(apply_nonlin=softmax_helper_dim1, batch_dice=batch_dice,
                                                      do_bg=False, smooth=1e-5, ddp=ddp)

            def forward(self, out, target):
                return self.ce(out, target[:, 0].long()) + self.dc(out, target)

        loss = _DCandCE(self.configuration_manager.batch_dice, self.is_ddp)
        return DeepSupervisionWrapper(loss, deep_supervision_weights(len(self._get_deep_supervision_scales())))

    # -- lifecycle ------------------------------------------------------------------------------
    def initialize(self):
        self.network = self.build_network_architecture(self.plans_manager, self.dataset_json,
                                                       self.configuration_manager, self.num_input_channels,
                                                       self.enable_deep_supervision).to(self.device)
        self.loss = self._build_loss()
        # nnU-Net's configure_optimizers: SGD(initial_lr, weight_decay, momentum 0.99, nesterov).  ClipSGD is that torch.optim.SGD with
        # the step (and train_step's gradient clip) on the library's step-glue kernels for float32 GPU parameters; anything else —
        # a CPU network in the tests — runs torch's own implementation inside it
        from .optim import ClipSGD
        self.optimizer = ClipSGD(self.network.parameters(), self.initial_lr, weight_decay=self.weight_decay,
                                 momentum=self.momentum, nesterov=True)
        return self

    def configure_rotation_dummyDA_mirroring_and_inital_patch_size(self):
        """(rotation, dummy-2d flag, initial patch size, mirror axes) — only the mirror axes matter to the plug-ins."""
        patch = tuple(self.configuration_manager.patch_size)
        return None, False, patch, tuple(range(len(patch)))

    def predict_logits(self, image: torch.Tensor, tile_step_size: float = 0.5, batch_size: int = 8,
                       autocast_dtype=None) -> torch.Tensor:
        """Sliding-window logits of one pre-processed case (C, *spatial), with this trainer's mirroring policy."""
        from .inference import predict_sliding_window
        self.configure_rotation_dummyDA_mirroring_and_inital_patch_size()   # lets *_NoMirroring veto the TTA
        return predict_sliding_window(self.network, image.to(self.device), self.configuration_manager.patch_size,
                                      tile_step_size, True, self.inference_allowed_mirroring_axes, batch_size,
                                      autocast_dtype)

    def train_step(self, data: torch.Tensor, target: List[torch.Tensor]) -> torch.Tensor:
        """forward -> deep-supervision loss -> backward -> clip -> SGD, like nnU-Net's train_step."""
        self.optimizer.zero_grad(set_to_none=True)
        loss = self.loss(self.network(data), target)
        loss.backward()
        if hasattr(self.optimizer, "clip_and_step"):
            self.optimizer.clip_and_step(12)
        else:
            torch.nn.utils.clip_grad_norm_(self.network.parameters(), 12)
            self.optimizer.step()
        return loss.detach()


class GraphedTrainStep:
    """One whole training step (zero_grad -> forward -> loss -> backward -> clip -> SGD) captured into a single hipGraph and
    replayed: the ~1700 kernel launches of a cfg-2 step become one, which takes the host out of the small stages (the s4 / s5
    graph blocks are a few microseconds of arithmetic behind ~100 launches each) and closes the idle gaps between kernels.

    ``step_fn`` must read its batch from tensors that live across replays (copy each new batch into them with ``copy_``) and
    must not synchronise with the host: no ``.item()``, no stochastic dilation (``torch.rand`` on the CPU); the (B)TI losses'
    target validation is switched to its deferred, device-side form when ``loss`` is given (see :meth:`check`).  ``warmup`` eager steps run first on a side stream (MIOpen's find, lazy momentum buffers, the library's
    one-time attribute calls); then one step is captured.  ``__call__`` replays it and returns the captured loss tensor.
    Same numbers as the eager step (same kernels, same order)."""

    def __init__(self, step_fn, warmup: int = 3, network: Optional[torch.nn.Module] = None, loss: Optional[torch.nn.Module] = None):
        from . import _lib
        from .loss.bti_loss import BTI_Loss
        if network is not None:
            assert_capturable(network)
        # (B)TI losses validate their targets with a host read per call (the reference's CrossEntropyLoss raises there,
        # bti_loss.py:141): inside a captured step the check runs on the device instead — a bad target turns the loss NaN and is
        # counted; check() raises the IndexError afterwards
        self._deferred = [m for m in (loss.modules() if loss is not None else ()) if isinstance(m, BTI_Loss) and m.validate_targets is True]
        for m in self._deferred:
            m.validate_targets = "deferred"
        _lib.lib().nextou_profile_enable(0)        # event records do not belong inside a captured graph
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(max(int(warmup), 1)):
                    step_fn()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            import torch.distributed as dist
            self.graph = torch.cuda.CUDAGraph()
            # With a process group up, RCCL's watchdog THREAD polls the events of in-flight collectives (hipEventQuery) whenever it likes;
            # under the default "global" capture mode such a call from another thread while this one captures is an error — and it is
            # raised inside the watchdog, which terminates the process (seen once in three runs of the world-size-1 averaged step:
            # "operation not permitted when stream is capturing").  "thread_local" restricts the check to the capturing thread.
            mode = "thread_local" if (dist.is_available() and dist.is_initialized()) else "global"
            with torch.cuda.graph(self.graph, capture_error_mode=mode):
                self.loss = step_fn()
        except BaseException:
            # a caller that falls back to the eager step (bench.py --graph auto) gets the reference's eager IndexError back: nobody
            # would ever call check() on a step that was not captured
            for m in self._deferred:
                m.validate_targets = True
            self._deferred = []
            raise

    def __call__(self):
        self.graph.replay()
        return self.loss

    def check(self) -> None:
        """Raise what the eager step would have raised on the way (one host synchronisation): out-of-range (B)TI targets."""
        for m in self._deferred:
            m.check_targets()


class SplitGraphedTrainStep:
    """The averaged (N > 1) training step as TWO hipGraphs around eager collectives (round 6): graph 1 = zero_grad -> forward -> loss ->
    backward with the gradient hooks only FILLING the flat buckets (``BucketedGradientAverager.defer_collectives``); then the bucket
    all-reduces, launched eagerly on RCCL's stream and joined by stream waits; graph 2 = 1 / world scale -> clip -> SGD.  No collective is ever
    captured, so PyTorch's RCCL watchdog thread never meets an event "last recorded in a capturing stream" (the abort that a captured averaged
    step can end in), and ~1 400 kernel launches per step still become two.  The price against :class:`GraphedTrainStep`
    with captured collectives is the overlap of the all-reduces with backward: ~1.4-3 ms of ring time for cfg 2's 122.7 MB of gradients,
    against the 6-8 ms the eager step loses to launch gaps (profiles/r06_cfg2_step_kernel_trace.md: 174.5 ms wall for 168.3 ms of kernels).

    ``part1()`` returns the loss tensor; ``between()`` launches and joins the collectives; ``part2()`` finishes the step.  All three must be
    free of host synchronisation (the averager's warm-up steps have cached its one host read).  Works with any backend: between the graphs
    the collectives are ordinary eager calls (gloo's host-side reductions included)."""

    def __init__(self, part1, between, part2, warmup: int = 2, network: Optional[torch.nn.Module] = None, loss: Optional[torch.nn.Module] = None):
        from . import _lib
        from .loss.bti_loss import BTI_Loss
        import torch.distributed as dist
        if network is not None:
            assert_capturable(network)
        self._deferred = [m for m in (loss.modules() if loss is not None else ()) if isinstance(m, BTI_Loss) and m.validate_targets is True]
        for m in self._deferred:
            m.validate_targets = "deferred"
        _lib.lib().nextou_profile_enable(0)
        self.between = between
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(max(int(warmup), 1)):
                    part1()
                    between()
                    part2()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            # thread_local: the watchdog thread may query the events of the (finished) eager collectives whenever it likes; only this thread's
            # calls are checked against the capture
            mode = "thread_local" if (dist.is_available() and dist.is_initialized()) else "global"
            self.graph1 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph1, capture_error_mode=mode):
                self.loss = part1()
            between()                      # real collectives on whatever the buckets hold: keeps the averager's bookkeeping in step order
            self.graph2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph2, pool=self.graph1.pool(), capture_error_mode=mode):
                part2()
            torch.cuda.synchronize()
        except BaseException:
            for m in self._deferred:
                m.validate_targets = True
            self._deferred = []
            raise

    def __call__(self):
        self.graph1.replay()
        self.between()
        self.graph2.replay()
        return self.loss

    def check(self) -> None:
        for m in self._deferred:
            m.check_targets()


def assert_capturable(network: torch.nn.Module) -> None:
    """Raise unless ``network``'s forward is the same computation on every replay of a captured step: a graph block whose
    kNN dilation is > 1 with ``stochastic`` set draws ``torch.rand`` on the host per call (reference torch_edge.py:126-136)
    — captured once, the draw would be frozen for every replay (and its ``randperm(...).to(device)`` is a pageable
    host-to-device copy, illegal during capture).  Every constructible NexToU has dilation 1, where the draw cannot change
    the result and is never made; set ``epsilon = 0`` / ``stochastic = False`` on the blocks to capture anything else."""
    from .network_architecture.torch_edge import DenseDilatedKnnGraph, reference_rng_stream
    for name, m in network.named_modules():
        if isinstance(m, DenseDilatedKnnGraph) and m.stochastic and m.epsilon > 0 and m.training and \
                (m.dilation > 1 or reference_rng_stream()):
            raise RuntimeError("GraphedTrainStep: %s draws its stochastic dilation on the host every call (dilation %d, epsilon %g); "
                               "a captured step would freeze the draw" % (name, m.dilation, m.epsilon))


def deep_supervision_weights(n_scales: int) -> np.ndarray:
    """1/2^i, last scale 0, normalised (reference nnUNetTrainer_NexToU_BTI_Synapse.py:23-27)."""
    w = np.array([1 / (2 ** i) for i in range(n_scales)])
    w[-1] = 0
    return w / w.sum()


def downsample_targets(target: torch.Tensor, outputs: Sequence[torch.Tensor]) -> List[torch.Tensor]:
    """Nearest-neighbour label pyramids matching the deep-supervision heads (what nnU-Net's
    DownsampleSegForDSTransform2 hands the loss)."""
    out = []
    for o in outputs:
        if tuple(o.shape[2:]) == tuple(target.shape[2:]):
            out.append(target)
        else:
            out.append(F.interpolate(target.float(), size=o.shape[2:], mode="nearest").to(target.dtype))
    return out


def synthetic_batch(configuration_manager, num_input_channels, num_classes, batch_size, device, seed=1234,
                    blob_labels=False):
    """i.i.d. N(0,1) images (nnU-Net z-scores its inputs) and integer label maps, generated on the
    device (SURVEY.md §8d 'synthetic inputs')."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    shape = [batch_size, num_input_channels] + list(configuration_manager.patch_size)
    data = torch.randn(shape, generator=g, device=device, dtype=torch.float32)
    lab_shape = [batch_size, 1] + list(configuration_manager.patch_size)
    if blob_labels:
        target = _blob_label_volume(lab_shape, num_classes, device, g)
    else:
        target = torch.randint(0, num_classes, lab_shape, generator=g, device=device).float()
    return data, target


def _blob_label_volume(shape, num_classes, device, g, n_seeds=40):
    """Nearest-seed Voronoi label blobs (BTCV-style, SURVEY.md §8d cfg 4): i.i.d. labels would make
    ~93 % of the voxels critical, which is unrepresentative for the BTI loss."""
    b, spatial = shape[0], shape[2:]
    dim = len(spatial)
    out = torch.zeros(shape, device=device)
    axes = [torch.arange(s, device=device, dtype=torch.float32) / s for s in spatial]
    grid = torch.stack(torch.meshgrid(*axes, indexing="ij"), -1).reshape(-1, dim)
    for i in range(b):
        pts = torch.rand((n_seeds, dim), generator=g, device=device)
        cls = torch.randint(1, num_classes, (n_seeds,), generator=g, device=device).float()
        nearest = torch.cdist(grid, pts).argmin(1)
        lab = cls[nearest]
        lab[((grid - 0.5).abs().max(1).values > 0.45)] = 0
        out[i, 0] = lab.reshape(spatial)
    return out
