"""Generates tests/golden/*.npz by RUNNING THE REFERENCE (PengchengShi1220/NexToU @ /root/reference).

Run once in the build container:  python tests/golden/make_golden.py
The reference's Python never leaves that container; only inputs/outputs (data) are committed.
/root/reference does not exist on the GPU box, so nothing at test time imports this script's
reference loader.

Import shims: the reference's architecture files import themselves through
``nnunetv2.…`` package paths and need two un-vendored third-party packages:
  * ``timm.models.layers.DropPath`` — symbol only (drop_path == 0 everywhere);
  * ``dynamic_network_architectures`` — StackedConvBlocks & helpers: provided by this repo's own
    restatement (nextou_amd/network_architecture/conv_blocks.py).  The plain conv stages of the
    model-level fixtures are therefore restated on both sides ("parity unpinned" for them); the
    graph blocks, losses and position tables are the reference's own code.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

import formula  # noqa: E402
from nextou_amd.network_architecture import conv_blocks  # noqa: E402  (own restatement, see docstring)

assert not conv_blocks.HAVE_DYNAMIC_NETWORK_ARCHITECTURES


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference():
    """-> namespace with the reference modules (torch_edge, torch_nn, pos_embed, encdec, nextou, bti)."""
    def pkg(name):
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
        return m

    timm = pkg("timm"); pkg("timm.models"); layers = pkg("timm.models.layers")
    layers.DropPath = type("DropPath", (nn.Identity,), {})
    timm.models = sys.modules["timm.models"]
    pkg("dynamic_network_architectures"); pkg("dynamic_network_architectures.building_blocks")
    scb = pkg("dynamic_network_architectures.building_blocks.simple_conv_blocks")
    scb.StackedConvBlocks = conv_blocks.StackedConvBlocks
    hlp = pkg("dynamic_network_architectures.building_blocks.helper")
    for f in ("convert_conv_op_to_dim", "convert_dim_to_conv_op", "get_matching_batchnorm",
              "get_matching_convtransp", "maybe_convert_scalar_to_list"):
        setattr(hlp, f, getattr(conv_blocks, f))
    hlp.get_matching_pool_op = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError())
    prefix = "nnunetv2.training.nnUNetTrainer.variants.network_architecture"
    parts = prefix.split(".")
    for i in range(1, len(parts) + 1):
        pkg(".".join(parts[:i]))
    ns = types.SimpleNamespace()
    arch = os.path.join(REF, "network_architecture")
    ns.torch_nn = _load(os.path.join(arch, "torch_nn.py"), prefix + ".torch_nn")
    ns.torch_edge = _load(os.path.join(arch, "torch_edge.py"), prefix + ".torch_edge")
    ns.pos_embed = _load(os.path.join(arch, "pos_embed.py"), prefix + ".pos_embed")
    ns.encdec = _load(os.path.join(arch, "NexToU_Encoder_Decoder.py"), prefix + ".NexToU_Encoder_Decoder")
    ns.nextou = _load(os.path.join(arch, "NexToU.py"), prefix + ".NexToU")
    ns.bti = _load(os.path.join(REF, "loss", "bti_loss.py"), "ref_bti_loss")
    ns.ti = _load(os.path.join(REF, "loss", "ti_loss.py"), "ref_ti_loss")
    # compound losses: the reference imports Dice / CE / softmax helper from the un-vendored nnunetv2 — provided by
    # this repo's restatement (nextou_amd/loss/nnunet_losses.py) on BOTH sides ("parity unpinned" for them); the
    # ignore-label / weighting glue and the (B)TI term are the reference's own code
    from nextou_amd.loss import nnunet_losses
    assert not nnunet_losses.HAVE_NNUNET
    for name in ("nnunetv2.training.loss", "nnunetv2.utilities"):
        pkg(name)
    dice = pkg("nnunetv2.training.loss.dice")
    dice.SoftDiceLoss, dice.MemoryEfficientSoftDiceLoss = nnunet_losses.SoftDiceLoss, nnunet_losses.MemoryEfficientSoftDiceLoss
    pkg("nnunetv2.training.loss.robust_ce_loss").RobustCrossEntropyLoss = nnunet_losses.RobustCrossEntropyLoss
    pkg("nnunetv2.utilities.helpers").softmax_helper_dim1 = nnunet_losses.softmax_helper_dim1
    sys.modules["nnunetv2.training.loss.bti_loss"] = ns.bti
    sys.modules["nnunetv2.training.loss.ti_loss"] = ns.ti
    ns.compound_bti = _load(os.path.join(REF, "loss", "compound_bti_loss.py"), "ref_compound_bti_loss")
    ns.compound_ti = _load(os.path.join(REF, "loss", "compound_ti_loss.py"), "ref_compound_ti_loss")
    return ns


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print("%-28s %8.1f KB  %s" % (name, os.path.getsize(path) / 1024, sorted(arrays)))


def kth_gap(ref, x, y, relpos, k):
    """gap between the k-th and (k+1)-th smallest reference distance of every row (fp64 recompute)."""
    xn = torch.nn.functional.normalize(x, p=2.0, dim=1).double()
    yn = xn if y is None else torch.nn.functional.normalize(y, p=2.0, dim=1).double()
    d = (xn * xn).sum(1).unsqueeze(2) - 2 * torch.einsum("bcn,bcm->bnm", xn, yn) + (yn * yn).sum(1).unsqueeze(1)
    if relpos is not None:
        d = d + relpos.double()
    s = torch.sort(d, dim=2).values
    if k >= s.shape[2]:
        return torch.full(s.shape[:2], float("inf"))
    return (s[:, :, k] - s[:, :, k - 1]).float()


# ------------------------------------------------------------------------------------------------
def g_knn(ref):
    cases = [  # name, B, C, N, M(None=self), k, d, with_relpos
        ("g1_self_a", 3, 12, 64, None, 9, 1, False), ("g1_self_a_rp", 3, 12, 64, None, 9, 1, True),
        ("g1_self_dil", 3, 12, 64, None, 4, 2, False), ("g1_self_dil_rp", 3, 12, 64, None, 4, 2, True),
        ("g1_window", 2, 132, 168, None, 7, 1, True),
        ("g2_xy", 2, 24, 256, 32, 14, 1, True), ("g2_xy_norp", 2, 24, 256, 32, 14, 1, False),
        ("g3_chunked", 1, 6, 10050, None, 4, 1, False),
    ]
    for name, B, C, N, M, k, d, with_rp in cases:
        x = formula.gaussian(name + ".x", (B, C, N, 1))
        y = None if M is None else formula.gaussian(name + ".y", (B, C, M, 1))
        rp = formula.gaussian(name + ".rp", (1, N, M or N), scale=0.05) if with_rp else None
        graph = ref.torch_edge.DenseDilatedKnnGraph(k, d, stochastic=False, epsilon=0.0).eval()
        edge = graph(x, y, rp)
        assert edge.shape == (2, B, N, k)
        full = ref.torch_edge.DenseDilatedKnnGraph(k * d, 1, stochastic=False, epsilon=0.0).eval()(x, y, rp)
        gap = kth_gap(ref, x.squeeze(-1), None if y is None else y.squeeze(-1), rp, k * d)
        save(name, x=x.numpy(), y=np.zeros(0) if y is None else y.numpy(),
             relpos=np.zeros(0) if rp is None else rp.numpy(), k=k, dilation=d,
             edge_index=edge.numpy().astype(np.int32), nn_full=full[0].numpy().astype(np.int32),
             kth_gap=gap.numpy())


def g_distance(ref):
    x = formula.gaussian("g_dist.x", (2, 40, 10))
    y = formula.gaussian("g_dist.y", (2, 24, 10))
    save("g_distance", x=x.numpy(), y=y.numpy(),
         pairwise=ref.torch_edge.pairwise_distance(x).numpy(),
         part=ref.torch_edge.part_pairwise_distance(x, 7, 19).numpy(),
         xy=ref.torch_edge.xy_pairwise_distance(x, y).numpy(),
         knn_unnormalised=ref.torch_edge.dense_knn_matrix(x.transpose(2, 1).unsqueeze(-1).contiguous(), 5)
         .numpy().astype(np.int32),
         xy_knn_unnormalised=ref.torch_edge.xy_dense_knn_matrix(
             x.transpose(2, 1).unsqueeze(-1).contiguous(), y.transpose(2, 1).unsqueeze(-1).contiguous(), 5)
         .numpy().astype(np.int32))


def g_mrconv(ref):
    """G4: batched_index_select + the pre-conv MRConv tensor + gradients (self and xy)."""
    out = {}
    for tag, M in (("self", None), ("xy", 20)):
        B, C, N, k = 2, 12, 48, 5
        x = formula.gaussian("g4.%s.x" % tag, (B, C, N, 1)).requires_grad_(True)
        y = None if M is None else formula.gaussian("g4.%s.y" % tag, (B, C, M, 1)).requires_grad_(True)
        rng = np.random.Generator(np.random.PCG64(7))
        nn_idx = torch.from_numpy(np.stack([np.stack([rng.permutation(M or N)[:k] for _ in range(N)])
                                            for _ in range(B)]).astype(np.int64))
        center = torch.arange(N).view(1, N, 1).expand(B, N, k)
        edge = torch.stack((nn_idx, center), 0)
        x_i = ref.torch_nn.batched_index_select(x, edge[1])
        x_j = ref.torch_nn.batched_index_select(x if y is None else y, edge[0])
        mr, _ = torch.max(x_j - x_i, -1, keepdim=True)
        pre = torch.cat([x.unsqueeze(2), mr.unsqueeze(2)], dim=2).reshape(B, 2 * C, N, 1)
        gout = formula.gaussian("g4.%s.g" % tag, pre.shape)
        grads = torch.autograd.grad(pre, [x] if y is None else [x, y], gout)
        out.update({tag + "_x": x.detach().numpy(), tag + "_idx": nn_idx.numpy().astype(np.int32),
                    tag + "_gather": x_j.detach().numpy(), tag + "_pre": pre.detach().numpy(),
                    tag + "_gout": gout.numpy(), tag + "_dx": grads[0].numpy()})
        if y is not None:
            out.update({tag + "_y": y.detach().numpy(), tag + "_dy": grads[1].numpy()})
    # a non-trivial centre index through the public MRConv.forward
    mr = ref.encdec.MRConv(12, 24, 'leakyrelu', 'instance', True, nn.Conv3d, None)
    formula.fill_module_(mr, seed=3)
    x = formula.gaussian("g4.pub.x", (2, 12, 48, 1))
    rng = np.random.Generator(np.random.PCG64(11))
    edge = torch.from_numpy(rng.integers(0, 48, size=(2, 2, 48, 5)).astype(np.int64))
    out.update(pub_x=x.numpy(), pub_edge=edge.numpy().astype(np.int32), pub_out=mr(x, edge).detach().numpy())
    save("g4_mrconv", **out)


def g_pos_embed(ref):
    """G6: sin-cos tables and final (negated, interpolated) relative_pos of graph blocks."""
    out = dict(rel2d_8_4=ref.pos_embed.get_2d_relative_pos_embed(8, 4),
               rel3d_12_3=ref.pos_embed.get_3d_relative_pos_embed(12, 3),
               sincos2d_8_3=ref.pos_embed.get_2d_sincos_pos_embed(8, 3),
               sincos3d_12_2=ref.pos_embed.get_3d_sincos_pos_embed(12, 2))
    opt = dict(norm_op=nn.BatchNorm3d, norm_op_kwargs={'eps': 1e-5, 'affine': True}, dropout_op=None)
    swin = ref.encdec.SwinGrapher(12, (4, 8, 8), 4, 1, 'mr', 'leakyrelu', 'instance', True, False, 0.2, 1, n=32,
                                  relative_pos=True, conv_op=nn.Conv3d, window_size=(2, 4, 4),
                                  shift_size=[1, 2, 2], **opt)
    out["swin_c12_n32_r1"] = swin.relative_pos.numpy()
    swin = ref.encdec.SwinGrapher(132, (4, 7, 6), 7, 1, 'mr', 'leakyrelu', 'instance', True, False, 0.2, 1,
                                  n=168, relative_pos=True, conv_op=nn.Conv3d, window_size=(4, 7, 6),
                                  shift_size=[2, 3, 3], **opt)
    out["swin_c132_n168_r1"] = swin.relative_pos.numpy()
    pool = ref.encdec.PoolGrapher(12, (4, 8, 8), 4, 1, 'mr', 'leakyrelu', 'instance', True, False, 0.2, 2, n=256,
                                  relative_pos=True, conv_op=nn.Conv3d, img_min_shape=(2, 4, 4), **opt)
    out["pool_c12_n256_r2"] = pool.relative_pos.numpy()
    pool2d = ref.encdec.PoolGrapher(8, (16, 16), 4, 1, 'mr', 'leakyrelu', 'instance', True, False, 0.2, 2, n=256,
                                    relative_pos=True, conv_op=nn.Conv2d, norm_op=nn.BatchNorm2d,
                                    norm_op_kwargs={'eps': 1e-5, 'affine': True}, dropout_op=None,
                                    img_min_shape=(4, 4))
    out["pool2d_c8_n256_r2"] = pool2d.relative_pos.numpy()
    save("g6_pos_embed", **out)


class Recorder:
    """Records, in call order, every kNN result and every real max-pool arg-max of a reference model."""

    def __init__(self, model, ref):
        self.entries = []
        self.handles = []
        for m in model.modules():
            if isinstance(m, ref.torch_edge.DenseDilatedKnnGraph):
                self.handles.append(m.register_forward_hook(
                    lambda mod, inp, out: self.entries.append(out[0].to(torch.int32).clone())))
            if isinstance(m, (nn.MaxPool2d, nn.MaxPool3d)) and m.return_indices:
                ks = m.kernel_size if isinstance(m.kernel_size, (list, tuple)) else [m.kernel_size]
                if any(int(k) != 1 for k in ks):
                    self.handles.append(m.register_forward_hook(
                        lambda mod, inp, out: self.entries.append(out[1].clone())))

    def close(self):
        for h in self.handles:
            h.remove()


class Replayer:
    """Teacher-forces a reference model: kNN results and real max-pool arg-max locations are
    replaced, in call order, by the recorded entries (forward hooks may return a new output)."""

    def __init__(self, model, ref, entries):
        self.entries, self.cursor, self.handles = entries, 0, []

        def knn_hook(mod, inp, out):
            nn_idx = self.entries[self.cursor].to(torch.int64)
            self.cursor += 1
            return torch.stack((nn_idx, out[1]), 0)

        def pool_hook(mod, inp, out):
            idx = self.entries[self.cursor]
            self.cursor += 1
            x = inp[0]
            vals = x.flatten(2).gather(2, idx.flatten(2)).reshape(idx.shape)
            return vals, idx

        for m in model.modules():
            if isinstance(m, ref.torch_edge.DenseDilatedKnnGraph):
                self.handles.append(m.register_forward_hook(knn_hook))
            if isinstance(m, (nn.MaxPool2d, nn.MaxPool3d)) and m.return_indices:
                ks = m.kernel_size if isinstance(m.kernel_size, (list, tuple)) else [m.kernel_size]
                if any(int(k) != 1 for k in ks):
                    self.handles.append(m.register_forward_hook(pool_hook))

    def close(self):
        for h in self.handles:
            h.remove()


def g_blocks(ref):
    """G5: PoolGrapher (pooled r=4 stage; un-pooled r=1 stage) and SwinGrapher, fwd + input grad,
    train and eval mode, with the recorded kNN / arg-max decisions for teacher forcing."""
    kw = dict(conv_op=nn.Conv3d, norm_op=nn.BatchNorm3d, norm_op_kwargs={'eps': 1e-5, 'affine': True},
              dropout_op=None)
    specs = {
        # img (8,16,32)=4096 > 64*32 -> pool (2,2,2) -> N=512, r=2 -> M=64
        "pool_pooled": lambda: ref.encdec.PoolGrapher(12, (8, 16, 32), 4, 1, 'mr', 'leakyrelu', 'instance', True,
                                                      True, 0.2, 2, n=4096, relative_pos=True,
                                                      img_min_shape=(2, 4, 4), **kw),
        "pool_plain": lambda: ref.encdec.PoolGrapher(12, (4, 8, 8), 6, 1, 'mr', 'leakyrelu', 'instance', True,
                                                     True, 0.2, 1, n=256, relative_pos=True,
                                                     img_min_shape=(2, 4, 4), **kw),
        "swin": lambda: ref.encdec.SwinGrapher(12, (4, 8, 8), 4, 1, 'mr', 'leakyrelu', 'instance', True, True,
                                               0.2, 1, n=32, relative_pos=True, window_size=(2, 4, 4),
                                               shift_size=[1, 2, 2], **kw),
    }
    shapes = {"pool_pooled": (2, 12, 8, 16, 32), "pool_plain": (2, 12, 4, 8, 8), "swin": (2, 12, 4, 8, 8)}
    out = {}
    for name, make in specs.items():
        blk = make()
        formula.fill_module_(blk, seed=5)
        for mode in ("train", "eval"):
            blk.train(mode == "train")
            # fresh running stats each time so the fixture does not depend on call history
            formula.fill_module_(blk, seed=5)
            x = formula.gaussian("g5.%s.x" % name, shapes[name]).requires_grad_(True)
            rec = Recorder(blk, ref)
            y = blk(x)
            rec.close()
            g = formula.gaussian("g5.%s.g" % name, y.shape)
            (dx,) = torch.autograd.grad(y, x, g)
            out["%s_%s_out" % (name, mode)] = y.detach().numpy()
            out["%s_%s_dx" % (name, mode)] = dx.numpy()
            for i, e in enumerate(rec.entries):
                out["%s_%s_tape%d" % (name, mode, i)] = e.numpy()
    save("g5_blocks", **out)


def g_bti(ref):
    """G7: critical map + loss + logit gradient for the shipped interaction lists."""
    def tens(lists):
        if not lists:
            return lists
        if isinstance(lists[0], list):
            return [tens(s) for s in lists]
        return torch.tensor(lists)

    synapse = [[[1, 3, 5, 7, 8, 11, 13], [2, 4, 6, 9, 10, 12]], [[1, 3, 11, 13], [5, 7, 8]], [[1, 3], [11, 13]],
               [1, 3], [11, 13], [[5, 8], [7]], [5, 8], [[4, 6, 10], [2, 9, 12]], [[4, 6], [10]], [4, 6],
               [[9, 12], [2]], [9, 12]]
    ica = [[[7, 9, 11, 12, 14, 15, 16, 17, 18], [1, 2, 3, 4, 5, 6, 8, 10, 13]], [[7, 9, 11, 12], [14, 15, 16, 17, 18]],
           [[7, 9], [11, 12]], [7, 9], [11, 12], [[14, 15], [16, 17, 18]], [14, 15], [[16, 17], [18]], [16, 17],
           [[3, 8, 10, 13], [1, 2, 4, 5, 6]], [[3, 10], [8, 13]], [3, 10], [8, 13], [[1, 6], [2, 4, 5]], [1, 6],
           [[2, 4], [5]], [2, 4]]
    cases = [  # name, dim, conn, classes, inclusion, exclusion, spatial
        ("synapse26", 3, 26, 14, [], synapse, (12, 20, 18)),
        ("synapse6", 3, 6, 14, [], synapse, (12, 20, 18)),
        ("ica26", 3, 26, 19, [], ica, (10, 12, 14)),
        ("ravir8", 2, 8, 3, [], [[1, 2]], (24, 28)),
        ("ravir4", 2, 4, 3, [], [[1, 2]], (24, 28)),
        ("incl26", 3, 26, 5, [[1, 2], [[3], [4]]], [[1, 3]], (8, 10, 12)),
    ]
    out = {}
    for name, dim, conn, L, inc, exc, sp in cases:
        loss = ref.bti.BTI_Loss(dim=dim, connectivity=conn, inclusion=tens(inc), exclusion=tens(exc), min_thick=1)
        logits = formula.gaussian("g7.%s.logits" % name, (2, L) + sp, scale=2.0)
        # make the arg-max spatially coherent: add a smooth per-class bias from blob labels
        for b in range(2):
            lab = torch.from_numpy(formula.blob_labels(sp, L, n_seeds=12, seed=100 + b))
            logits[b].scatter_add_(0, lab.unsqueeze(0), torch.full((1,) + sp, 3.0))
        target = torch.from_numpy(np.stack([formula.blob_labels(sp, L, n_seeds=12, seed=200 + b)
                                            for b in range(2)])).unsqueeze(1).float()
        logits.requires_grad_(True)
        value = loss(logits, target)
        (grad,) = torch.autograd.grad(value, logits)
        with torch.no_grad():
            P = torch.argmax(torch.softmax(logits, 1), 1).unsqueeze(1).double()
            crit = loss.binary_topological_interaction_module(P)
        out.update({name + "_logits": logits.detach().numpy(), name + "_target": target.numpy().astype(np.uint8),
                    name + "_labels": P.squeeze(1).numpy().astype(np.uint8),
                    name + "_critical": crit.squeeze(1).numpy().astype(np.uint8),
                    name + "_loss": value.detach().numpy(), name + "_grad": grad.numpy()})
    save("g7_bti", **out)


def g_near_ties(ref):
    """G7d: the label map ``argmax(softmax(x, 1), 1)`` (bti_loss.py:131-133, with the module's own ``apply_nonlin``) on logits whose two
    largest entries are closer than float32 softmax can tell apart — the first-index-on-equal-softmax behaviour the kernels restate."""
    loss = ref.bti.BTI_Loss(dim=3, connectivity=26, inclusion=[], exclusion=[[torch.tensor(1), torch.tensor(2)]], min_thick=1)
    logits, gap = formula.near_tie_logits("g7d.near_ties")
    with torch.no_grad():
        labels = torch.argmax(loss.apply_nonlin(logits), dim=1)
    plain = logits.argmax(1)
    print("   near ties: %d voxels, reference labels differ from argmax(logits) on %d; gap <= 2^-25: %d" % (
        labels.numel(), int((labels != plain).sum()), int((gap <= 2.0 ** -25).sum())))
    # logits are formula.near_tie_logits("g7d.near_ties") on both sides: not stored
    save("g7d_near_ties", labels=labels.numpy().astype(np.uint8))


def g_ti(ref):
    """G7b: the all-pairs TI loss of loss/ti_loss.py (scalar labels, `P == label`): the 78 pairs of the 13 foreground
    classes the `*_TI` trainers build (nnUNetTrainer_NexToU_TI.py:10-13,48), a 2-D 8-connected case, one inclusion."""
    from itertools import combinations

    def tens(lists):
        if not lists:
            return lists
        if isinstance(lists[0], list):
            return [tens(s) for s in lists]
        return torch.tensor(lists)

    cases = [  # name, dim, conn, classes, inclusion, exclusion, spatial
        ("ti78_26", 3, 26, 14, [], [list(c) for c in combinations(range(1, 14), 2)], (12, 20, 18)),
        ("ti78_6", 3, 6, 14, [], [list(c) for c in combinations(range(1, 14), 2)], (12, 20, 18)),
        ("ti10_8", 2, 8, 6, [], [list(c) for c in combinations(range(1, 6), 2)], (24, 28)),
        ("ti_incl", 3, 26, 5, [[1, 2]], [[3, 4]], (8, 10, 12)),
    ]
    out = {}
    for name, dim, conn, L, inc, exc, sp in cases:
        loss = ref.ti.TI_Loss(dim=dim, connectivity=conn, inclusion=tens(inc), exclusion=tens(exc), min_thick=1)
        logits, target = formula.coherent_logits("g7b.%s" % name, L, sp)
        logits.requires_grad_(True)
        value = loss(logits, target)
        (grad,) = torch.autograd.grad(value, logits)
        with torch.no_grad():
            P = torch.argmax(torch.softmax(logits, 1), 1).unsqueeze(1).double()
            crit = loss.topological_interaction_module(P)
        # logits / target are formula.coherent_logits("g7b.<name>", L, sp) on both sides: not stored
        out.update({name + "_labels": P.squeeze(1).numpy().astype(np.uint8),
                    name + "_critical": crit.squeeze(1).numpy().astype(np.uint8),
                    name + "_loss": value.detach().numpy(), name + "_grad": grad.numpy(),
                    name + "_n_interactions": len(loss.interaction_list)})
    save("g7b_ti", **out)


def g_compound(ref):
    """G7c: DC_and_CE_and_BTI_Loss / DC_and_CE_and_TI_Loss (compound_bti_loss.py:33-61): value and logit gradient with
    and without an ignore label.  Dice / CE are this repo's restatement of nnU-Net's on both sides (see load_reference)."""
    from itertools import combinations
    from nextou_amd.loss.nnunet_losses import MemoryEfficientSoftDiceLoss
    synapse = [[[1, 3, 5, 7, 8, 11, 13], [2, 4, 6, 9, 10, 12]], [[1, 3, 11, 13], [5, 7, 8]], [[1, 3], [11, 13]],
               [1, 3], [11, 13], [[5, 8], [7]], [5, 8], [[4, 6, 10], [2, 9, 12]], [[4, 6], [10]], [4, 6],
               [[9, 12], [2]], [9, 12]]

    def tens(lists):
        if not lists:
            return lists
        if isinstance(lists[0], list):
            return [tens(s) for s in lists]
        return torch.tensor(lists)

    dice = {'batch_dice': False, 'smooth': 1e-5, 'do_bg': False, 'ddp': False}
    out = {}
    for name, cls, exc, ignore, batch_dice in (
            ("bti", ref.compound_bti.DC_and_CE_and_BTI_Loss, synapse, None, False),
            ("bti_ignore", ref.compound_bti.DC_and_CE_and_BTI_Loss, synapse, 13, False),
            ("bti_batchdice", ref.compound_bti.DC_and_CE_and_BTI_Loss, synapse, None, True),
            ("ti", ref.compound_ti.DC_and_CE_and_TI_Loss, [list(c) for c in combinations(range(1, 14), 2)], None, False)):
        L, sp = 14, (12, 20, 18)
        ti = {'dim': 3, 'connectivity': 26, 'inclusion': [], 'exclusion': tens(exc), 'min_thick': 1}
        loss = cls(dict(dice, batch_dice=batch_dice), {}, ti, weight_ce=1, weight_dice=1, weight_ti=1e-6,
                   ignore_label=ignore, dice_class=MemoryEfficientSoftDiceLoss)
        logits, target = formula.coherent_logits("g7c.%s" % name, L, sp)
        if ignore is not None:
            # a slab of ignored voxels.  The ignore label must be < L here: with nnU-Net's usual encoding (ignore ==
            # number of classes) the reference's (B)TI term raises IndexError("Target 14 is out of bounds") from its
            # float64 CrossEntropyLoss (bti_loss.py:141) as soon as one ignored voxel is present
            target[:, :, :3] = float(ignore)
        logits.requires_grad_(True)
        value = loss(logits, target)
        (grad,) = torch.autograd.grad(value, logits)
        out.update({name + "_loss": value.detach().numpy(), name + "_grad": grad.numpy(),
                    name + "_ignore": -1 if ignore is None else ignore, name + "_batch_dice": int(batch_dice)})
    save("g7c_compound", **out)


def g_ffn(ref):
    """G5b: FFN (NexToU_Encoder_Decoder.py:368-390) forward + input gradient, train and eval."""
    out = {}
    for mode in ("train", "eval"):
        ffn = ref.encdec.FFN(12, 48, act='leakyrelu', drop_path=0.0, conv_op=nn.Conv3d, norm_op=nn.BatchNorm3d,
                             norm_op_kwargs={'eps': 1e-5, 'affine': True})
        formula.fill_module_(ffn, seed=6)
        ffn.train(mode == "train")
        x = formula.gaussian("g5b.ffn.x", (2, 12, 4, 8, 8)).requires_grad_(True)
        y = ffn(x)
        g = formula.gaussian("g5b.ffn.g", y.shape)
        (dx,) = torch.autograd.grad(y, x, g)
        out["ffn_%s_out" % mode] = y.detach().numpy()
        out["ffn_%s_dx" % mode] = dx.numpy()
        if mode == "train":
            out["ffn_train_running_mean_fc1"] = ffn.fc1[1].running_mean.numpy().copy()
            out["ffn_train_running_var_fc1"] = ffn.fc1[1].running_var.numpy().copy()
    save("g5b_ffn", **out)


def build_ref_model(ref, cfg):
    return ref.nextou.NexToU(
        input_channels=cfg["in_ch"], patch_size=cfg["patch"], n_stages=len(cfg["kernels"]),
        features_per_stage=cfg["features"], conv_op=cfg["conv_op"], kernel_sizes=cfg["kernels"],
        strides=cfg["strides"], n_conv_per_stage=2, num_classes=cfg["classes"], n_conv_per_stage_decoder=2,
        conv_bias=True, norm_op=cfg["norm_op"], norm_op_kwargs={'eps': 1e-5, 'affine': True}, dropout_op=None,
        dropout_op_kwargs=None, nonlin=nn.LeakyReLU, nonlin_kwargs={'inplace': True}, deep_supervision=True)


TINY_2D = dict(in_ch=1, patch=[64, 64], features=[8, 16, 32, 64, 64], conv_op=nn.Conv2d, norm_op=nn.BatchNorm2d,
               kernels=[[3, 3]] * 5, strides=[[1, 1]] + [[2, 2]] * 4, classes=3)
TINY_3D = dict(in_ch=1, patch=[32, 128, 128], features=[6, 12, 24, 48, 48, 48], conv_op=nn.Conv3d,
               norm_op=nn.BatchNorm3d, kernels=[[1, 3, 3]] + [[3, 3, 3]] * 5,
               strides=[[1, 1, 1], [1, 2, 2]] + [[2, 2, 2]] * 4, classes=4)


def g_models(ref):
    """G8: tiny full models, train-mode BN, teacher-forcing tape + logits (+ self-noise floor)."""
    for name, cfg, B in (("g8_tiny2d", TINY_2D, 2), ("g8_tiny3d", TINY_3D, 1)):
        model = build_ref_model(ref, cfg)
        formula.fill_module_(model, seed=1)
        model.train()
        x = formula.gaussian(name + ".x", [B, cfg["in_ch"]] + cfg["patch"])
        rec = Recorder(model, ref)
        with torch.no_grad():
            outs = model(x)
        rec.close()
        # self-noise floor: the reference against ITSELF under 1e-7 relative input noise (below one
        # fp32 ulp), teacher-forced with the decisions of the clean run (SURVEY §7 hard part 0):
        # the smallest |dlogit| any other fp32 implementation of the dense stages can be held to.
        rep = Replayer(model, ref, rec.entries)
        with torch.no_grad():
            noisy = model(x * (1 + 1e-7 * formula.gaussian(name + ".noise", x.shape)))
        rep.close()
        assert rep.cursor == len(rec.entries)
        floor = max(float((a - b).abs().max()) for a, b in zip(outs, noisy))
        absmax = max(float(o.abs().max()) for o in outs)
        print("   %s: max|logit| %.2f, self-noise floor (1e-7 input noise, teacher-forced) %.3e" % (name, absmax, floor))
        # the same reference model with every convolution computed in float64 (formula.convs_in_float64), teacher-forced
        # with the decisions of the fp32 run: the "equal convolution arithmetic" target of the GPU parity test
        rep = Replayer(model, ref, rec.entries)
        with torch.no_grad(), formula.convs_in_float64():
            outs64 = model(x)
        rep.close()
        assert rep.cursor == len(rec.entries)
        print("   %s: fp64-conv vs fp32-conv reference logits differ by %.3e" % (
            name, max(float((a - b).abs().max()) for a, b in zip(outs, outs64))))
        arrays = {"x": x.numpy(), "n_tape": len(rec.entries), "self_noise_floor": floor, "logit_absmax": absmax}
        for i, o in enumerate(outs64):
            if o.numel() <= 300_000:
                arrays["f64conv_logits%d" % i] = o.numpy()
            else:
                arrays["f64conv_logits%d_sample" % i] = o.reshape(-1)[::97].numpy()
        for i, e in enumerate(rec.entries):
            arrays["tape%d" % i] = e.numpy()
        for i, o in enumerate(outs):
            if o.numel() <= 300_000:
                arrays["logits%d" % i] = o.numpy()
            else:  # full-resolution heads: strided sample + checksum-friendly statistics
                flat = o.reshape(-1)
                arrays["logits%d_sample" % i] = flat[::97].numpy()
                arrays["logits%d_shape" % i] = np.asarray(o.shape)
                arrays["logits%d_absmax" % i] = float(o.abs().max())
        arrays["n_heads"] = len(outs)
        arrays["state_keys"] = np.asarray(sorted(model.state_dict().keys()))
        save(name, **arrays)


def g_state_dict(ref):
    """G8s: the state_dict the REFERENCE model emits (tiny 3-D model of g8_tiny3d, same formula weights, the reference's own
    `relative_pos` parameters, every alias key — `decoder.encoder.*`, `all_modules.*`): keys in state_dict order, one array per distinct
    storage, and for every key the index of its storage.  tests: load_state_dict(strict=True) into an un-initialised nextou_amd model
    must reproduce g8_tiny3d's logits."""
    model = build_ref_model(ref, TINY_3D)
    formula.fill_module_(model, seed=1)
    sd = model.state_dict()
    keys, owner, store = [], [], {}
    arrays = {}
    for k, v in sd.items():
        ident = (v.data_ptr(), tuple(v.shape), tuple(v.stride()), str(v.dtype))
        if ident not in store:
            store[ident] = len(store)
            arrays["t%d" % store[ident]] = v.detach().cpu().numpy()
        keys.append(k)
        owner.append(store[ident])
    arrays["keys"] = np.asarray(keys)
    arrays["storage_of_key"] = np.asarray(owner, dtype=np.int32)
    n_values = sum(int(np.prod(a.shape)) for k, a in arrays.items() if k.startswith("t"))
    print("   %d keys, %d distinct tensors, %d values; relative_pos keys: %d" % (
        len(keys), len(store), n_values, sum(k.endswith("relative_pos") for k in keys)))
    save("g8_tiny3d_state", **arrays)


def g_config_table(ref):
    """G9: GNN hyper-parameters the reference derives for cfg 1 / cfg 2 / cfg 5,
    read from live module attributes of (cheap) stand-alone block factories."""
    rows = {}
    cfgs = {
        "cfg2": (nn.Conv3d, [64, 224, 192], [[1, 1, 1], [1, 2, 2]] + [[2, 2, 2]] * 4),
        "cfg5": (nn.Conv3d, [96, 256, 256], [[1, 1, 1], [1, 2, 2]] + [[2, 2, 2]] * 4),
        "cfg1": (nn.Conv2d, [512, 512], [[1, 1]] + [[2, 2]] * 6),
    }
    for name, (conv_op, patch, strides) in cfgs.items():
        dim = len(patch)
        shapes = [tuple(patch)]
        for st in strides[1:]:
            shapes.append(tuple(s // p for s, p in zip(shapes[-1], st)))
        n_stages = len(strides)
        opt = ref.encdec.OptInit(pool_op_kernel_sizes_len=n_stages)
        opt.img_min_shape = shapes[-1]
        opt.n_size_list = [int(np.prod(s)) for s in shapes]
        norm = nn.BatchNorm3d if dim == 3 else nn.BatchNorm2d
        table = []
        for s in range(n_stages - 4, n_stages):
            i = s - (n_stages - 4)
            # channel count 12 keeps the position tables tiny; k / r / pool do not depend on it
            n_pts = int(np.prod(shapes[s]))
            if n_pts > 20000:   # building these blocks would allocate the multi-GB position tables
                table.append([s, -1, -1, -1, -1, -1])
                continue
            pool = ref.encdec.PoolGNNBlocks(12, shapes[s], i, n_stages - 4, opt=opt, conv_op=conv_op, norm_op=norm,
                                            norm_op_kwargs={'eps': 1e-5, 'affine': True}, dropout_op=None)
            swin = ref.encdec.SwinGNNBlocks(12, shapes[s], i, opt=opt, conv_op=conv_op, norm_op=norm,
                                            norm_op_kwargs={'eps': 1e-5, 'affine': True}, dropout_op=None)
            pg, sg = pool.blocks[0][0], swin.blocks[0][0]
            table.append([s, pg.graph_conv.k, pg.graph_conv.r, int(np.prod(pg.pool_size)), pg.n, sg.graph_conv.k])
        rows[name] = np.asarray(table)
    save("g9_config_table", **rows)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref = load_reference()
    only = set(sys.argv[1:])
    for fn in (g_knn, g_distance, g_mrconv, g_pos_embed, g_blocks, g_ffn, g_bti, g_near_ties, g_ti, g_compound, g_models, g_state_dict, g_config_table):
        if only and fn.__name__ not in only:
            continue
        print("==", fn.__name__)
        fn(ref)


if __name__ == "__main__":
    main()
