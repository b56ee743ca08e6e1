"""Grouped execution engine: runs the SAME layer of all E experts as one HIP launch.

The reference loops over experts in Python and calls ~60 cuDNN/cuBLAS kernels per expert
(``model/moe.py:141``).  Here the experts are folded into the image index (image n of the NHWC
activation belongs to expert ``n // B``) and every layer -- conv, BatchNorm pass, pool, ECA, MLP
GEMM -- is issued once for all experts.  Activations stay on the device in NHWC bf16 (or f32)
between launches; parameters stay in the reference's per-expert f32 tensors (the state_dict
contract) and are repacked to the grouped kernel layouts by one small launch per layer.

Backward is driven by a tape of closures recorded during forward (no torch autograd inside the
path); the whole network is ONE ``torch.autograd.Function`` towards the outside, so ``loss.backward()``,
``clip_grad_norm_`` and optimizers of the reference trainer work unchanged.  Gradients are written
into one flat f32 arena in backward order, which is also the data-parallel all-reduce buffer
(bucketed RCCL all-reduce launched while earlier layers are still in backward).
"""
import itertools
import os

import torch
import torch.distributed as dist

from . import hip, ops
from .model import blocks as B
from .parallel import BucketedAllReduce

F32 = torch.float32


def r16(c):
    return (c + 15) // 16 * 16


def r64(c):
    return (c + 63) // 64 * 64


class Var:
    """A device activation [N,H,W,ld] (+ channel window) with its gradient slot."""
    __slots__ = ("t", "coff", "c", "g", "needs_grad", "act", "drop_p", "base", "gap_part", "bn_src", "bn_part", "f8", "bn_defer",
                 "bn_fused", "pending_bn")

    def __init__(self, t, c=None, coff=0, needs_grad=False, base=None):
        self.t, self.coff, self.c = t, coff, (c if c is not None else t.shape[-1])
        self.g, self.needs_grad, self.act, self.drop_p, self.base = None, needs_grad, hip.ACT_NONE, 0.0, base
        self.gap_part = None                 # (part [N,nparts,C], nparts): channel sums left by the pass that wrote t (_bn, want_gap)
        self.bn_src = None                   # (z, layer, coef [4,E,C], rpe): t = relu(BatchNorm(z)) in train mode (_bn)
        self.bn_part = None                  # (part, nparts): that BatchNorm's backward reductions, left by the consumer's dgrad
        self.f8 = None                       # the same activation as e4m3(t * in_scale) bytes (fp8 policy: _bn, want_f8)
        self.bn_defer = False                # t feeds a BatchNorm and the ONLY consumer of dL/dt applies that BatchNorm's backward on load
        self.bn_fused = None                 # (g, coef, c1, c2) left by _bn_bwd for that consumer instead of dL/dt (round 4: _stem_in_bwd)
        self.pending_bn = None               # (coef [4,E,C], rpe): t is the PRE-activation z of a BatchNorm + ReLU that its one consumer
                                             # applies on load (PMOE_RES_INBN) or engine_punet._materialize writes out (untaped PU-Net forward)

    @property
    def grad(self):
        return self.base.g if self.base is not None else self.g

    def set_grad(self, g):
        if (self.base if self.base is not None else self).bn_part is not None:
            # a PMOE_RES_DBN data gradient already masked this gradient with the ReLU decision and reduced it for the
            # BatchNorm backward (_dgrad_with_bn_reduce): it assumed this activation has exactly one consumer.  A second
            # contribution would be added unmasked and would be missing from the reductions -- silently wrong dgamma / dbeta / dz
            raise RuntimeError("relu(BatchNorm(z)) whose backward reductions came out of its consumer's data gradient "
                               "(PMOE_RES_DBN) must have exactly one consumer; a second gradient contribution arrived "
                               "(set PMOE_BN_REDUCE_IN_DGRAD=0 for graphs that fan this activation out)")
        if self.base is not None:
            self.base.g = g
        else:
            self.g = g

    def window(self, coff, c):
        v = Var(self.t, c, coff, self.needs_grad, base=self if self.base is None else self.base)
        return v


class GroupedConv:
    """One conv / linear layer of all experts: packed operands + launch helpers."""

    def __init__(self, eng, name, weights, biases, cin, cout, ks, stride, pad):
        self.eng, self.name = eng, name
        self.weights, self.biases = weights, biases
        self.cin, self.cout, self.ks, self.stride, self.pad = cin, cout, ks, stride, pad
        self.cinp, self.coutp, self.cout_st = r16(cin), r64(cout), r16(cout)
        self.dg_rows, self.dg_red = r64(cin), r16(cout)
        self.taps = ks * ks
        self.w_fwd = self.w_dg = self.bias_packed = None
        self.need_dgrad = True
        self.fp8_ok = False          # set for the ResNet layer1-4 convolutions (the fp8 policy's selection)
        self.w_f8 = self.wscale = self.oscale = None

    @property
    def fp8(self):
        """this layer's forward runs on e4m3 operands (BASELINE config 5: engine.fp8 + bf16 activations).  Round 3: the
        policy selects the dense 3x3 stride-1 convolutions with whole 128-channel chunks (ResNet layer2-4: 46 % of the forward
        MACs) -- the launches the block-scaled fp8 matrix instruction serves at twice the bf16 rate; everything else stays bf16."""
        return (self.fp8_ok and self.eng.fp8 and self.ks == 3 and self.stride == 1 and self.cin % 128 == 0
                and self.cin >= self.eng.fp8_min_cin)

    def alloc(self, dtype, dev):
        E = self.eng.E
        if self.fp8 and dtype == torch.bfloat16:
            self.w_f8 = torch.empty(E, self.coutp, self.taps, self.cinp, dtype=torch.uint8, device=dev)
            self.wscale = torch.empty(E, self.coutp, dtype=F32, device=dev)
            self.oscale = torch.empty(E, self.coutp, dtype=F32, device=dev)
            self.w_fwd = None        # an fp8 layer's forward operand is w_f8: no (never packed) bf16 copy of the bank
        else:
            self.w_f8 = self.wscale = self.oscale = None
            self.w_fwd = torch.empty(E, self.coutp, self.taps, self.cinp, dtype=dtype, device=dev)
        self._alloc_key = (dtype, dev, self.w_f8 is not None)
        self.w_dg = torch.empty(E, self.dg_rows, self.taps, self.dg_red, dtype=dtype, device=dev) if self.need_dgrad else None
        if self.biases is not None:
            self.bias_packed = torch.empty(E, self.coutp, dtype=F32, device=dev)

    def pack(self, wtab, btab):
        E = self.eng.E
        if self.w_f8 is not None:
            # e4m3 forward operand + per-channel scales; the data-gradient operand holds the exactly dequantised weights
            ops.pack_conv_weights_fp8(wtab, self.w_f8, self.w_dg, self.wscale, self.oscale, self.eng.fp8_in_scale, E,
                                      self.cout, self.cin, self.ks, self.coutp, self.cinp, self.dg_rows, self.dg_red)
            return
        ops.pack_conv_weights(wtab, self.w_fwd, self.w_dg, E, self.cout, self.cin, self.ks, self.coutp, self.cinp,
                              self.dg_rows, self.dg_red, self.w_fwd.dtype)
        if self.biases is not None:
            ops.pack_bias(btab, self.bias_packed, E, self.cout, self.coutp)

    @property
    def trainable(self):
        return any(p.requires_grad for p in self.weights) or (
            self.biases is not None and any(p.requires_grad for p in self.biases))


class GroupedBN:
    def __init__(self, name, mods):
        self.name, self.mods = name, mods
        self.C = mods[0].num_features
        self.eps, self.momentum = mods[0].eps, mods[0].momentum

    @property
    def trainable(self):
        return any(m.weight.requires_grad or m.bias.requires_grad for m in self.mods)


class GroupedECA:
    def __init__(self, name, mods):
        self.name, self.mods = name, mods
        self.k = mods[0].conv.kernel_size
        self.creal = mods[0].channels

    @property
    def trainable(self):
        return any(m.conv.weight.requires_grad for m in self.mods)


class ExpertGroupEngine:
    """Executes ``experts`` (list of pmoe_amd.model.moe.BaseExpert[Alt]) as one grouped network."""

    def __init__(self, experts, alt=False, shared_k=0):
        """``shared_k`` > 0: MixtureOfExpertsShared (moe.py:180-233) -- ``experts`` is the one module that owns the
        shared trunk (group of one) and its head emits ``shared_k`` mixture components per sample."""
        self.experts = list(experts)
        self.E = len(self.experts)
        self.alt = alt
        self.shared = shared_k > 0
        self.K = shared_k if self.shared else self.E       # mixture components per sample
        if self.shared and (self.E != 1 or alt or 5 * self.K > 64):
            raise ValueError("shared-trunk mixture: one trunk, plain alpha head, n_experts <= 12")
        self.dp_group = None          # torch.distributed process group for gradient all-reduce (None = WORLD)
        self.dp_enabled = False       # set by pmoe_amd.parallel-aware callers (bench.py, enable_data_parallel)
        self.dp_buckets = 6
        self.dp_always = False        # issue the collectives even in a one-rank group (RCCL path on a single GPU)
        self._built_for = None
        self._seed_counter = itertools.count(1)
        self.fuse_conv_stats = True
        self.fuse_stem_tail = True
        self.fold_stem_input = True
        # weight gradients on a side HIP stream, overlapped with the BatchNorm backward passes and the next data gradient
        # (+2 % step throughput at the headline shape).  Off by default: co-running kernels stretch each other, which
        # makes per-kernel durations (bench.py roofline, rocprofv3 traces) meaningless; PMOE_OVERLAP_WGRAD=1 turns it on
        self.overlap_wgrad = os.environ.get("PMOE_OVERLAP_WGRAD", "0") == "1"
        self.fold_bn_eval = True      # inference: eval-mode BatchNorm folded into the conv weights / epilogue
        self.pooled_stem_bwd = True   # stem-tail BatchNorm reductions from the pooled tensors (train mode)
        self.fold_eca_gate = True     # ECA gate folded into per-image conv weights (no gated activation in memory)
        # residual-block BatchNorm backward: the reduce pass stores the ReLU-masked gradient, the apply pass reads it (7 tensor
        # passes per block output instead of 8).  PMOE_BN_MASK_IN_REDUCE=0: A/B switch
        self.bn_mask_in_reduce = os.environ.get("PMOE_BN_MASK_IN_REDUCE", "1") != "0"
        # round 3: the reductions of a BatchNorm+ReLU's backward come out of the data-gradient epilogue of the convolution that
        # consumed its output (PMOE_RES_DBN: one extra read of z there instead of a pass over dy and z).  bf16, LDS-DMA kernels
        self.bn_reduce_in_dgrad = os.environ.get("PMOE_BN_REDUCE_IN_DGRAD", "1") != "0"
        # stem: the BatchNorm+ReLU pass that writes a1 also leaves the ECA block's per-image channel sums (no GAP pass over a1)
        self.fuse_bn_gap = os.environ.get("PMOE_FUSE_BN_GAP", "1") != "0"
        # round 4: the stem's first BatchNorm backward applied on load by conv1's per-image filter gradient (no dz1 tensor)
        self.stem_bn_fuse = os.environ.get("PMOE_STEM_BN_FUSE", "1") != "0"
        # round 4: weight + bias gradient of the expert MLP layers in one small launch each (PMOE_MLP_WGRAD=0: the generic
        # weight-gradient kernel + fold + column-sum chain of rounds 1-3)
        self.mlp_wgrad_fused = os.environ.get("PMOE_MLP_WGRAD", "1") != "0"
        # BASELINE config 5: e4m3 weights + e4m3 activations on the fp8 matrix cores for the layer1-4 forward convolutions
        # (policy: include/pmoe_hip.h, pmoe_pack_conv_weights_fp8).  fp8_min_cin: smallest input-channel count that takes it
        self.fp8 = False
        self.fp8_in_scale = 16.0
        self.fp8_min_cin = int(os.environ.get("PMOE_FP8_MIN_CIN", "64"))
        self.raw_alpha = False        # lone BaseExpert.forward: the gate kernel returns alpha itself instead of softmax(alpha)
        self.debug_grads = None       # dict -> backward stores the gradient entering every BatchNorm (tests/experiments/probe_layers.py)
        # dict -> forward keeps a reference to every tensor that carries a DISCRETE decision of the network: the outputs of
        # ReLU layers (BatchNorm+ReLU passes and ReLU GEMM epilogues, by layer name) and the max-pool's winning taps
        # ("maxpool").  tests/forced_masks.py hands them to its float64 checker so that both sides differentiate the SAME
        # piecewise-linear function (two f32 evaluations disagree on the sign of pre-activations within ~1e-7 of zero)
        self.debug_acts = None
        self._collect()

    # ------------------------------------------------------------------ structure
    def _collect(self):
        ex = self.experts
        self.params = []              # ordered list of (kind, layer, [E params]) in FORWARD order

        def conv(name, mods, ks=None):
            m0 = mods[0]
            if isinstance(m0, B.Linear):
                layer = GroupedConv(self, name, [m.weight for m in mods],
                                    [m.bias for m in mods] if m0.bias is not None else None,
                                    m0.in_features, m0.out_features, 1, 1, 0)
            else:
                layer = GroupedConv(self, name, [m.weight for m in mods],
                                    [m.bias for m in mods] if getattr(m0, "bias", None) is not None else None,
                                    m0.in_channels, m0.out_channels, m0.kernel_size, m0.stride, m0.padding)
            self.params.append(("w", layer, layer.weights))
            if layer.biases is not None:
                self.params.append(("b", layer, layer.biases))
            return layer

        def bn(name, mods):
            layer = GroupedBN(name, mods)
            self.params.append(("gamma", layer, [m.weight for m in mods]))
            self.params.append(("beta", layer, [m.bias for m in mods]))
            return layer

        def eca(name, mods):
            layer = GroupedECA(name, mods)
            self.params.append(("eca", layer, [m.conv.weight for m in mods]))
            return layer

        def mlp(name, seqs):
            spec = seqs[0].spec
            for s in seqs:
                if s.spec != spec:
                    raise ValueError("experts of one mixture must share MLP configuration")
            lins = [[m for m in s if isinstance(m, B.Linear)] for s in seqs]
            bns = [[m for m in s if isinstance(m, B.BatchNorm1d)] for s in seqs]
            layers, bn_layers = [], []
            for i in range(len(lins[0])):             # forward order: Linear i, then its BatchNorm1d (basics.py:30-34)
                layers.append(conv(f"{name}.{i}", [l[i] for l in lins]))
                if spec["bn"] and i < len(lins[0]) - 1:
                    bn_layers.append(bn(f"{name}.bn{i}", [b_[i] for b_ in bns]))
            return dict(layers=layers, bns=bn_layers if spec["bn"] else None, act=hip.ACT_BY_NAME[spec["act"]],
                        l_act=spec["l_act"], dropout=spec["dropout"])

        self._mk = dict(conv=conv, bn=bn, eca=eca, mlp=mlp)
        self._collect_network(ex)
        del self._mk
        if self.conv1 is not None:
            self.conv1.need_dgrad = self._stem_needs_dgrad()
        self.all_convs = [p[1] for p in self.params if p[0] == "w"]
        self.all_bns = [p[1] for p in self.params if p[0] == "gamma"]
        self.flat_params = [p for _, _, plist in self.params for p in plist]

    def _collect_network(self, ex):
        """layers in FORWARD order (the gradient arena is laid out in the reverse of this order)."""
        mlp = self._mk["mlp"]
        self.speed_enc = mlp("speed_encoder", [e.speed_encoder for e in ex])
        self.cmd_enc = mlp("command_encoder", [e.command_encoder for e in ex])
        self._collect_pre_backbone(ex)
        self._collect_backbone([e.backbone for e in ex])
        self._collect_heads(ex)

    def _collect_pre_backbone(self, ex):
        """hook: layers that run before the ResNet backbone (the frozen PU-Net of PUNetExpert)."""

    def _stem_needs_dgrad(self):
        return True

    def _collect_backbone(self, bbs):
        conv, bn, eca = self._mk["conv"], self._mk["bn"], self._mk["eca"]
        self.eca1 = eca("eca1", [b.conv1.layer1.eca1 for b in bbs])
        self.conv1 = conv("stem.conv1", [b.conv1.layer1.conv1[0] for b in bbs])
        self.bn_c1 = bn("stem.bn1", [b.conv1.layer1.conv1[1] for b in bbs])
        self.eca2 = eca("eca2", [b.conv1.layer2.eca2 for b in bbs])
        self.conv2 = conv("stem.conv2", [b.conv1.layer2.conv2[0] for b in bbs])
        self.bn_c2 = bn("stem.bn2", [b.conv1.layer2.conv2[1] for b in bbs])
        self.bn1 = bn("bn1", [b.bn1 for b in bbs])
        self.blocks = []
        for li in range(1, 5):
            seqs = [getattr(b, f"layer{li}") for b in bbs]
            for bi in range(len(seqs[0])):
                blks = [s[bi] for s in seqs]
                d = dict(conv1=conv(f"layer{li}.{bi}.conv1", [k.conv1 for k in blks]),
                         bn1=bn(f"layer{li}.{bi}.bn1", [k.bn1 for k in blks]),
                         conv2=conv(f"layer{li}.{bi}.conv2", [k.conv2 for k in blks]),
                         bn2=bn(f"layer{li}.{bi}.bn2", [k.bn2 for k in blks]), down=None)
                if blks[0].downsample is not None:
                    d["down"] = (conv(f"layer{li}.{bi}.down", [k.downsample[0] for k in blks]),
                                 bn(f"layer{li}.{bi}.downbn", [k.downsample[1] for k in blks]))
                    d["down"][0].fp8_ok = True
                d["conv1"].fp8_ok = d["conv2"].fp8_ok = True
                self.blocks.append(d)

    def _collect_heads(self, ex):
        conv, mlp = self._mk["conv"], self._mk["mlp"]
        self.speed_pred = mlp("speed_pred", [e.speed_pred for e in ex])
        self.action_feat = mlp("action_features", [e.action_features for e in ex])
        if self.alt:
            # BaseExpertAlt (moe.py:108-110): alpha = Sequential(Linear(1536,512), ReLU, Linear(512,1)) on features
            self.alpha_mlp = dict(layers=[conv("alpha.0", [e.alpha[0] for e in ex]), conv("alpha.2", [e.alpha[2] for e in ex])],
                                  act=hip.ACT_RELU, l_act=False, dropout=0.0)
            self.head = conv("action_pred", [e.action_pred for e in ex])
        else:
            # action_pred (4 rows) and alpha (1 row) read the same input: one GEMM with 5 output rows
            self.head = self._fused_head(ex)

    def _fused_head(self, ex):
        class _Cat:  # weights of action_pred (rows 0..3) and alpha (row 4) as one 5-row layer
            pass
        k = self.K if self.shared else 1      # shared trunk: rows 0..4K-1 action_pred, rows 4K..5K-1 alpha
        layer = GroupedConv(self, "head", None, None, ex[0].action_pred.in_features, 5 * k, 1, 1, 0)
        layer.parts = [([e.action_pred.weight for e in ex], [e.action_pred.bias for e in ex], 4 * k),
                       ([e.alpha.weight for e in ex], [e.alpha.bias for e in ex], k)]
        layer.weights = layer.parts[0][0] + layer.parts[1][0]
        layer.biases = layer.parts[0][1] + layer.parts[1][1]
        for (ws, bs, _), nm in zip(layer.parts, ("action_pred", "alpha")):
            self.params.append(("w_part", (layer, nm), ws))
            self.params.append(("b_part", (layer, nm), bs))
        return layer

    # ------------------------------------------------------------------ buffers
    def _ensure_built(self, dev, dtype):
        key = (str(dev), dtype, self.fp8, self.fp8_min_cin)
        if self._built_for == key:
            return
        for layer in self.all_convs + ([self.head] if getattr(self.head, "parts", None) else []):
            if getattr(layer, "_alloc_key", None) != (dtype, dev, bool(layer.fp8 and dtype == torch.bfloat16)):
                layer.alloc(dtype, dev)
        self._ptr_key = None
        self._packed_version = None
        self._built_for = key

    def _refresh_tables(self, dev):
        """One int64 device table of all parameter/buffer pointers (rebuilt only when storage moved)."""
        tensors = []
        index = {}

        def add(key, lst):
            index[key] = (len(tensors), len(lst))
            tensors.extend(lst)

        for kind, layer, plist in self.params:
            add((kind, id(layer) if not isinstance(layer, tuple) else (id(layer[0]), layer[1])), plist)
        for bnl in self.all_bns:
            add(("rm", id(bnl)), [m.running_mean for m in bnl.mods])
            add(("rv", id(bnl)), [m.running_var for m in bnl.mods])
        for key, lst in self._extra_tables():
            add(key, lst)
        ptrs = tuple(t.data_ptr() for t in tensors)
        if ptrs != self._ptr_key:
            for t in tensors:
                if t.device != dev or t.dtype != F32 or not t.is_contiguous():
                    raise RuntimeError("pmoe_amd: parameters/buffers must be contiguous float32 tensors on the "
                                       f"same device as the inputs ({dev}); got {t.dtype} on {t.device}")
            self._ptr_tab = torch.tensor(ptrs, dtype=torch.int64, device=dev)
            self._ptr_key = ptrs
            self._ptr_index = index
            self._packed_version = None

    def _extra_tables(self):
        """hook: additional (key, [tensors]) pointer-table rows (padded BatchNorm shadows of the PU-Net engine)."""
        return []

    def _tab(self, kind, layer):
        k = (kind, id(layer) if not isinstance(layer, tuple) else (id(layer[0]), layer[1]))
        s, n = self._ptr_index[k]
        return self._ptr_tab[s:s + n]

    def _pack_all(self):
        ver = sum(p._version for p in self.flat_params)
        if ver == self._packed_version:
            return
        for layer in self.all_convs:
            layer.pack(self._tab("w", layer), self._tab("b", layer) if layer.biases is not None else None)
        if getattr(self.head, "parts", None):
            h = self.head
            E = self.E
            # two packs into row windows of the fused 5-row head: rows 0..3 action_pred, row 4 alpha
            for (ws, bs, rows), nm, r0 in ((h.parts[0], "action_pred", 0), (h.parts[1], "alpha", h.parts[0][2])):
                self._pack_head_part(h, nm, rows, r0)
        self._packed_version = ver

    def _pack_head_part(self, h, nm, rows, r0):
        # The pack kernel writes a whole [coutp] panel, so pack each part into scratch and copy its rows.
        E = self.E
        scratch_f = torch.empty(E, 64, 1, h.cinp, dtype=h.w_fwd.dtype, device=h.w_fwd.device)
        scratch_d = torch.empty(E, h.dg_rows, 1, h.dg_red, dtype=h.w_fwd.dtype, device=h.w_fwd.device)
        ops.pack_conv_weights(self._tab("w_part", (h, nm)), scratch_f, scratch_d, E, rows, h.cin, 1, 64, h.cinp,
                              h.dg_rows, h.dg_red, h.w_fwd.dtype)
        if r0 == 0:
            h.w_fwd.zero_()
            h.w_dg.zero_()
            h.bias_packed.zero_()
        h.w_fwd[:, r0:r0 + rows] = scratch_f[:, :rows]
        h.w_dg[:, :, :, r0:r0 + rows] = scratch_d[:, :, :, :rows]
        sb = torch.empty(E, 64, dtype=F32, device=h.w_fwd.device)
        ops.pack_bias(self._tab("b_part", (h, nm)), sb, E, rows, 64)
        h.bias_packed[:, r0:r0 + rows] = sb[:, :rows]

    # ------------------------------------------------------------------ primitive ops (forward + tape)
    def _new(self, n, h, w, c, dtype=None):
        return torch.empty(n, h, w, c, dtype=dtype or self.dtype, device=self.dev)

    def _conv(self, x, layer, *, bias=True, act=hip.ACT_NONE, drop_p=0.0, out=None, out_coff=0, in_shared=False,
              want_stats=False, tape=True, in_bn=None):
        """``in_bn``: x is the pre-activation z of a BatchNorm + ReLU and in_bn its [4, E, C] coefficient block -- the launch applies
        them on load (PMOE_RES_INBN: untaped forward launches; the caller has asked pmoe_conv2d_plan)."""
        H, W = x.t.shape[1], x.t.shape[2]
        Ho = ops.conv_out_size(H, layer.ks, layer.stride, layer.pad)
        Wo = ops.conv_out_size(W, layer.ks, layer.stride, layer.pad)
        if out is None:
            o = Var(self._new(self.N, Ho, Wo, layer.cout_st), layer.cout_st, 0)
        else:
            o = out.window(out_coff, layer.cout_st)
        stats = None
        f8 = layer.w_f8 is not None
        if f8 and (bias is not False or act != hip.ACT_NONE):
            raise RuntimeError(f"{layer.name}: an fp8-policy layer has only its e4m3 forward operand (bias-free, no activation)")
        # e4m3 activations left by the producing BatchNorm pass: the block-scaled MFMA kernel (no conversion in the loader)
        xin = x.t
        if f8 and x.f8 is not None and x.coff == 0 and not in_shared:
            kwp = dict(cin=layer.cinp, cout=layer.cout_st, coutp=layer.coutp, ipe=self.B, ks=layer.ks, stride=layer.stride,
                       pad=layer.pad, out_coff=o.coff, out_scale=layer.oscale, in_scale=self.fp8_in_scale)
            if ops.conv2d(x.f8, layer.w_f8, o.t, plan_only=True, **kwp) == 8507:
                xin = x.f8
        in8 = xin is not x.t
        if want_stats:
            rows = ops.conv2d_stat_rows(self.N, H, W, Ho, Wo, layer.cinp, layer.cout_st, layer.coutp, self.B, layer.ks,
                                        layer.stride, layer.pad, self.dtype, w_fp8=f8, in_ld=xin.shape[-1],
                                        out_ld=o.t.shape[-1], in_shared=in_shared, in_fp8=in8)
            stats = torch.empty(rows, 2, layer.coutp, dtype=F32, device=self.dev)
        seed = (next(self._seed_counter) * 0x9E3779B1 + self.base_seed) & 0xFFFFFFFFFFFF if drop_p > 0 else 0
        flop = 2.0 * self.N * Ho * Wo * layer.cout * layer.cin * layer.taps
        ops.set_meta(flop=flop, name=layer.name)
        ops.conv2d(xin, layer.w_f8 if f8 else layer.w_fwd, o.t, cin=layer.cinp, cout=layer.cout_st, coutp=layer.coutp,
                   ipe=self.B, ks=layer.ks, stride=layer.stride, pad=layer.pad, in_shared=in_shared, in_coff=x.coff,
                   out_coff=o.coff, bias=layer.bias_packed if bias else None, act=act, drop_p=drop_p, seed=seed,
                   stats=stats, out_scale=layer.oscale if f8 else None, in_scale=self.fp8_in_scale,
                   **(dict(res_mode=hip.RES_INBN, bn_coef=in_bn) if in_bn is not None else {}))
        o.act, o.drop_p = act, drop_p
        if self.debug_acts is not None and act == hip.ACT_RELU:
            self.debug_acts[layer.name] = (o.t, o.coff, layer.cout)
        o.needs_grad = x.needs_grad or layer.trainable
        if out is not None and o.needs_grad:
            out.needs_grad = True
        if tape and self.taping and o.needs_grad:
            self.tape.append(lambda: self._conv_bwd(x, layer, o, in_shared, flop))
        return (o, stats) if want_stats else o

    def _conv_bwd(self, x, layer, o, in_shared, flop=0.0):
        """dy is the gradient w.r.t. the layer's PRE-activation output (the consumer's dgrad epilogue
        already applied act'); emits weight/bias gradients and, if needed, the input gradient."""
        dy = o.grad
        if dy is None:
            return
        E = self.E
        if layer.trainable and self.overlap_wgrad:
            # weight / bias gradients on a second HIP stream: they depend on nothing downstream, so the MFMA-bound
            # wgrad kernel overlaps the HBM-bound BatchNorm-backward passes and the next data gradient of the main stream
            main, side = torch.cuda.current_stream(), self._side_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                self._wgrad_block(x, layer, o, dy, in_shared, flop)
            x.t.record_stream(side)
            dy.record_stream(side)
        elif layer.trainable:
            self._wgrad_block(x, layer, o, dy, in_shared, flop)
        self._dgrad_block(x, layer, o, dy, flop)

    def _side_stream(self):
        st = self.__dict__.get("_side")
        if st is None or st.device != self.dev:
            st = self.__dict__["_side"] = torch.cuda.Stream(device=self.dev)
        return st

    def _wgrad_block(self, x, layer, o, dy, in_shared, flop):
        E = self.E
        if (self.mlp_wgrad_fused and layer.ks == 1 and self.dtype == torch.bfloat16 and dy.shape[1] == 1 and dy.shape[2] == 1
                and x.t.shape[1] == 1 and x.t.shape[2] == 1 and not hasattr(layer, "store_grads")):
            # round 4: a Linear layer of the expert MLPs -- weight AND bias gradient in one launch, written in the parameters'
            # own layout (csrc/gemm_skinny.hip mlp_wgrad_kernel); the fused 5-row head goes through a dense temporary
            parts = getattr(layer, "parts", None)
            ops.set_meta(flop=flop, name=layer.name)
            if parts is None:
                ops.mlp_wgrad(x.t, dy, self._grad_slot("w", layer), self._grad_slot("b", layer) if layer.biases is not None else None,
                              cin=layer.cinp, cout=layer.cout_st, cin_real=layer.cin, cout_real=layer.cout, ipe=self.B,
                              x_shared=in_shared, x_coff=x.coff, dy_coff=o.coff)
                return
            full = torch.empty(E, layer.cout, layer.cin, dtype=F32, device=self.dev)
            fb = torch.empty(E, layer.cout, dtype=F32, device=self.dev)
            ops.mlp_wgrad(x.t, dy, full, fb, cin=layer.cinp, cout=layer.cout_st, cin_real=layer.cin, cout_real=layer.cout,
                          ipe=self.B, x_shared=in_shared, x_coff=x.coff, dy_coff=o.coff)
            ra = parts[0][2]
            self._grad_slot("w_part", (layer, "action_pred")).view(E, ra, layer.cin).copy_(full[:, 0:ra])
            self._grad_slot("w_part", (layer, "alpha")).view(E, layer.cout - ra, layer.cin).copy_(full[:, ra:])
            self._grad_slot("b_part", (layer, "action_pred")).view(E, ra).copy_(fb[:, 0:ra])
            self._grad_slot("b_part", (layer, "alpha")).view(E, layer.cout - ra).copy_(fb[:, ra:])
            return
        ckw = 64 if self.dtype == torch.bfloat16 else 32
        cpw = (layer.cinp + ckw - 1) // ckw * ckw
        cow = (layer.cout_st + ckw - 1) // ckw * ckw
        ws = self._wgrad_ws(E * layer.taps * cow * cpw)      # overwritten by the launch (no atomics, nothing to zero)
        parts = getattr(layer, "parts", None)
        # plain layers: the launch's own fold writes the parameter-layout gradient (no separate unpack launch)
        direct = parts is None and not hasattr(layer, "store_grads")
        ops.set_meta(flop=flop, name=layer.name)
        d = ops.conv2d_wgrad(x.t, dy, ws, cin=layer.cinp, cout=layer.cout_st, cinp=cpw, coutp=cow, ipe=self.B,
                             ks=layer.ks, stride=layer.stride, pad=layer.pad, x_shared=in_shared, x_coff=x.coff,
                             dy_coff=o.coff, grads=self._grad_slot("w", layer) if direct else None, grads_cout=layer.cout,
                             grads_cin=layer.cin, defer_fold=direct)
        if direct:
            ops.conv2d_wgrad_fold(d)
        if hasattr(layer, "store_grads"):       # derived layouts (ConvTranspose2d as a 4*Cout-row 1x1 layer, engine_punet)
            layer.store_grads(self, ws, cow, cpw)
            return
        if parts is None:
            pass                                           # written by the launch (direct)
        else:
            full = torch.empty(E, layer.cout, layer.cin, dtype=F32, device=self.dev)
            ops.unpack_conv_wgrad(ws, full, E, layer.cout, layer.cin, 1, cow, cpw)
            ra = parts[0][2]
            self._grad_slot("w_part", (layer, "action_pred")).view(E, ra, layer.cin).copy_(full[:, 0:ra])
            self._grad_slot("w_part", (layer, "alpha")).view(E, layer.cout - ra, layer.cin).copy_(full[:, ra:])
        if layer.biases is not None:
            sums = self._colsum(dy, self.B * dy.shape[1] * dy.shape[2], layer.cout_st, o.coff)[:, :layer.cout]
            if parts is None:
                self._grad_slot("b", layer).view(E, layer.cout).copy_(sums)
            else:
                ra = parts[0][2]
                self._grad_slot("b_part", (layer, "action_pred")).view(E, ra).copy_(sums[:, 0:ra])
                self._grad_slot("b_part", (layer, "alpha")).view(E, layer.cout - ra).copy_(sums[:, ra:])

    def _colsum(self, t, rpe, C, coff=0):
        """per-expert column sums [E,C] (f32) of the channel window [coff, coff+C) of ``t``: partial rows over enough
        workgroups to stream at HBM rate, then fixed-order folds (bias gradients of conv layers with many pixels)."""
        E = self.E
        nparts = self._nparts(rpe)
        part = torch.empty(E, nparts, 2, C, dtype=F32, device=self.dev)
        ops.colstats(rpe, t, E, C, part, nparts, ld=t.shape[-1], coff=coff)
        if nparts > 1:
            part, nparts = self._fold_parts(part, nparts, 2 * C)
            one = torch.empty(E, 1, 2, C, dtype=F32, device=self.dev)
            ops.reduce_partials(part, one, E, nparts, 1, 2 * C)
            part = one
        return part[:, 0, 0]

    def _dgrad_block(self, x, layer, o, dy, flop):
        if x.needs_grad:
            prev = x.grad
            res, res_mode = None, hip.RES_NONE
            if x.act != hip.ACT_NONE:
                if prev is not None:
                    raise RuntimeError("activation output consumed twice: unsupported gradient accumulation")
                res, res_mode = x.t, hip.RES_OF_ACT[x.act]
            elif prev is not None:
                res, res_mode = prev, hip.RES_ADD       # second consumer: accumulate in the epilogue
            g = prev if prev is not None else torch.empty_like(x.t)
            kw = dict(cin=layer.dg_red, cout=layer.cinp, coutp=layer.dg_rows, ipe=self.B, ks=layer.ks, stride=1,
                      pad=layer.ks - 1 - layer.pad, dilate=(layer.stride == 2), in_coff=o.coff, out_coff=x.coff)
            if self._dgrad_with_bn_reduce(x, dy, layer.w_dg, g, kw, flop, layer.name):
                return
            ops.set_meta(flop=flop, name=layer.name + ":dgrad")
            ops.conv2d(dy, layer.w_dg, g, res=res, res_coff=x.coff, res_mode=res_mode,
                       drop_p=x.drop_p if res_mode >= hip.RES_DRELU else 0.0, **kw)
            x.set_grad(g)

    def _dgrad_with_bn_reduce(self, x, dy, w_dg, g, kw, flop, name, bias=None):
        """x = relu(BatchNorm(z)) with this conv as its only consumer: run the data gradient with PMOE_RES_DBN -- the epilogue
        reads z, masks the gradient with the recomputed ReLU decision and leaves sum(g), sum(g * xhat) per channel, i.e. the
        reduce pass of that BatchNorm's backward (_bn_bwd then only finalizes and applies).  False = not applicable here (other
        dtype / kernel / geometry): the caller runs the plain data gradient."""
        src = x.bn_src
        if (src is None or x.grad is not None or x.act != hip.ACT_NONE or x.base is not None or x.coff != 0
                or kw["ks"] != 3 or kw["dilate"] or x.t.dtype != torch.bfloat16):
            return False
        z, bnl, coef, rpe = src
        if bnl.C != kw["cout"] or bnl.C != kw["coutp"] or z.t.shape != x.t.shape:      # (the statistics rows are coutp wide)
            return False
        common = dict(res=z.t, res_mode=hip.RES_DBN, bn_coef=coef, bn_ipe=self.B, bias=bias, **kw)
        if ops.conv2d(dy, w_dg, g, plan_only=True, **common) not in (1107, 1117, 1247, 1257, 5007, 5017):
            return False
        n, h, w = dy.shape[0], dy.shape[1], dy.shape[2]
        rows = ops.conv2d_stat_rows(n, h, w, h, w, kw["cin"], kw["cout"], kw["coutp"], kw["ipe"], 3, 1, 1, self.dtype,
                                    in_ld=dy.shape[-1], out_ld=g.shape[-1])
        stats = torch.empty(rows, 2, kw["coutp"], dtype=F32, device=self.dev)
        ops.set_meta(flop=flop, name=name + ":dgrad+bnred")
        ops.conv2d(dy, w_dg, g, stats=stats, **common)
        x.set_grad(g)
        x.bn_part = (stats, rows // self.E)
        return True

    @staticmethod
    def _nparts(rpe):
        """partial-sum rows per expert of a streaming reduction: enough workgroups to keep HBM busy
        (E * nparts blocks of 256 threads), at least 256 rows each."""
        return min(1024, max(1, rpe // 256))

    def _fold_parts(self, part, nparts, width, cap=128):
        """deterministic second stage: [E][nparts][width] -> at most 128 rows before a finalize kernel (``cap``: what that
        kernel folds by itself -- the BatchNorm finalize kernels take 2048 rows over 16 partial lanes)."""
        if nparts <= cap:
            return part, nparts
        small = torch.empty(self.E, 128, width, dtype=F32, device=self.dev)
        ops.reduce_partials(part, small, self.E, nparts, 128, width)
        return small, 128

    def _bn_coeffs(self, layer, rpe, part=None, nparts=0, z=None, shiftc=None):
        """scale/shift/mean/invstd [E,C] of a BatchNorm: batch statistics from partial sums (train) or the
        running buffers (eval); updates the running statistics in train mode (momentum, unbiased var)."""
        E, C_ = self.E, layer.C
        # one block [mean | invstd | scale | shift][E][C]: the layout PMOE_RES_DBN's data-gradient epilogue reads
        coef = self._last_coef = torch.empty(4, E, C_, dtype=F32, device=self.dev)
        mean, invstd, scale, shift = coef[0], coef[1], coef[2], coef[3]
        if self.training:
            if part is None:
                nparts = self._nparts(rpe)
                part = torch.empty(E, nparts, 2, C_, dtype=F32, device=self.dev)
                shiftc = torch.empty(E, C_, dtype=F32, device=self.dev)
                ops.colstats(rpe, z.t, E, C_, part, nparts, shiftc=shiftc)
            part, nparts = self._fold_parts(part, nparts, 2 * C_, cap=2048)
            ops.bn_finalize(part, nparts, rpe, self._tab("gamma", layer), self._tab("beta", layer),
                            self._tab("rm", layer), self._tab("rv", layer), layer.momentum, layer.eps, True, scale,
                            shift, mean, invstd, E, C_, shiftc)
            self._bn_touched.append(layer)
        else:
            ops.bn_finalize(scale, 0, rpe, self._tab("gamma", layer), self._tab("beta", layer), self._tab("rm", layer),
                            self._tab("rv", layer), layer.momentum, layer.eps, False, scale, shift, mean, invstd, E, C_)
        return scale, shift, mean, invstd

    def _bn(self, z, layer, relu, res=None, stats=None, out=None, out_coff=0, want_gap=False, want_f8=False, pool_to=None):
        """y = [relu](bn(z) [+ res]); train mode: batch statistics (fused conv partials or a colstats pass).
        want_gap: the pass also leaves the per-image channel sums of y in ``y.gap_part`` (part, nparts) for the ECA block that
        follows (_eca_conv_folded), instead of a separate pass over y."""
        E, C_ = self.E, layer.C
        n, h, w, _ = z.t.shape
        rpe = self.B * h * w
        if stats is not None and stats.shape[2] != C_:
            raise RuntimeError("fused stats width mismatch")
        scale, shift, mean, invstd = self._bn_coeffs(layer, rpe, stats, stats.shape[0] // E if stats is not None else 0, z)
        # out: write into the channel window [out_coff, out_coff + C) of a wider buffer (U-Net skip concatenation)
        # taped: the window gets its OWN (dense) gradient slot -- its consumers' backward fills it (U-Net skip: the
        # max-pool backward adds the concatenation buffer's gradient window, engine_punet._maxpool2)
        if out is None:
            y = Var(torch.empty_like(z.t))
        elif self.taping:
            y = Var(out.t, C_, out_coff)
        else:
            y = out.window(out_coff, C_)
        ops.set_meta(name=layer.name, bytes=z.t.numel() * z.t.element_size() * (3 if res is not None else 2))
        if want_gap and res is None and out is None and self.fuse_bn_gap:
            nparts = self._gap_parts(h * w)
            part = torch.empty(n, nparts, C_, dtype=F32, device=self.dev)
            ops.bn_apply_gap(z.t, y.t, scale, shift, mean, part, nparts, self.B, relu)
            y.gap_part = (part, nparts)
        elif pool_to is not None and res is None:
            # round 4: the pass also leaves MaxPool2d(2, 2) of its output in ``pool_to`` (U-Net down blocks, engine_punet._unet_fwd)
            ops.bn_apply_pool2(z.t, y.t, pool_to, scale, shift, mean, self.B, E, C_, relu, y_coff=y.coff)
        else:
            if want_f8 and out is None and self.dtype == torch.bfloat16:
                y.f8 = torch.empty(y.t.shape, dtype=torch.uint8, device=self.dev)      # e4m3(y * in_scale) for the fp8 consumer
            ops.bn_apply(z.t, res.t if res is not None else None, y.t, scale, shift, mean, rpe, E, C_, relu, y_coff=y.coff,
                         y_fp8=y.f8, in_scale=self.fp8_in_scale)
        if self.debug_acts is not None and relu:
            self.debug_acts[layer.name] = (y.t, y.coff, C_)
        if (relu and res is None and out is None and self.taping and self.training and self.bn_reduce_in_dgrad
                and self.dtype == torch.bfloat16 and z.needs_grad):
            y.bn_src = (z, layer, self._last_coef, rpe)
        y.needs_grad = z.needs_grad or layer.trainable or (res is not None and res.needs_grad)
        if out is not None and y.needs_grad:
            out.needs_grad = True
        if self.taping and y.needs_grad:
            train = self.training
            self.tape.append(lambda: self._bn_bwd(z, layer, y, res, relu, scale, shift, mean, invstd, rpe, train))
        return y

    def _bn_bwd(self, z, layer, y, res, relu, scale, shift, mean, invstd, rpe, train):
        dy = y.grad
        if self.debug_grads is not None and dy is not None:
            self.debug_grads[layer.name] = dy.detach().clone()      # gradient w.r.t. this BN's (post-activation) output
        # without a residual the ReLU mask is a function of z alone: skip reading the saved output
        ysrc = y.t if (res is not None or not relu) else None
        if dy is None:
            return
        E, C_ = self.E, layer.C
        nb = z.t.numel() * z.t.element_size()
        if y.bn_part is not None:
            # the consumer's data gradient (PMOE_RES_DBN) already masked dy and reduced it: finalize + apply only
            part, nparts = y.bn_part
            if part.shape[2] != C_:
                raise RuntimeError("BatchNorm reductions from the data-gradient epilogue: channel count mismatch")
            part, nparts = self._fold_parts(part.view(E, nparts, 2 * C_), nparts, 2 * C_, cap=2048)
            c1, c2 = (torch.empty(E, C_, dtype=F32, device=self.dev) for _ in range(2))
            dgamma, dbeta, store = self._bn_grad_views(layer)
            ops.bn_bwd_finalize(part, nparts, rpe, dgamma, dbeta, c1, c2, E, C_)
            if store is not None:
                store()
            if z.needs_grad:
                if z.grad is not None:
                    raise RuntimeError("BN input consumed twice")
                if z.bn_defer:
                    # the only consumer of dz evaluates it on load from (g, z): no apply pass, dz is never written
                    z.bn_fused = (dy, y.bn_src[2], c1, c2)
                    z.set_grad(dy)
                    return
                dz = torch.empty_like(z.t)
                ops.set_meta(name=layer.name, bytes=nb * 3)
                ops.bn_bwd_apply(dy, None, z.t, mean, invstd, scale, shift, c1, c2, dz, None, rpe, E, C_, False)
                z.set_grad(dz)
            return
        nparts = self._nparts(rpe)
        part = torch.empty(E, nparts, 2, C_, dtype=F32, device=self.dev)
        want_res = res is not None and res.needs_grad
        # residual blocks: the reduce pass also stores the masked gradient (the residual branch needs it anyway), and the
        # apply pass then reads that instead of dy AND the saved output: one tensor pass less per block
        gm = torch.empty_like(z.t) if (want_res and relu and ysrc is not None and self.bn_mask_in_reduce) else None
        ops.set_meta(name=layer.name, bytes=nb * ((3 if ysrc is not None else 2) + (gm is not None)))
        ops.bn_bwd_reduce(dy, ysrc, z.t, mean, invstd, scale, shift, rpe, E, C_, relu, part, nparts, gmask=gm)
        part, nparts = self._fold_parts(part, nparts, 2 * C_, cap=2048)
        c1, c2 = (torch.empty(E, C_, dtype=F32, device=self.dev) for _ in range(2))
        dgamma, dbeta, store = self._bn_grad_views(layer)
        ops.bn_bwd_finalize(part, nparts, rpe, dgamma, dbeta, c1, c2, E, C_)
        if store is not None:
            store()
        if not train:          # eval-mode BN is an affine map: no batch-statistics terms
            c1.zero_()
            c2.zero_()
        dz = torch.empty_like(z.t) if z.needs_grad else None
        if gm is not None:
            if dz is not None:
                ops.set_meta(name=layer.name, bytes=nb * 3)
                ops.bn_bwd_apply(gm, None, z.t, mean, invstd, scale, shift, c1, c2, dz, None, rpe, E, C_, False)
        else:
            gm = torch.empty_like(z.t) if want_res else None
            if dz is None and gm is None:
                return
            if dz is None:
                dz = torch.empty_like(z.t)
            ops.set_meta(name=layer.name, bytes=nb * (2 + (ysrc is not None) + 1 + (gm is not None)))
            ops.bn_bwd_apply(dy, ysrc, z.t, mean, invstd, scale, shift, c1, c2, dz, gm, rpe, E, C_, relu)
        if self.debug_grads is not None:
            self.debug_grads[layer.name + ":dz"] = dz.detach().clone()
            self.debug_grads[layer.name + ":stats"] = (mean.clone(), invstd.clone(), c1.clone(), c2.clone())
        if z.needs_grad:
            if z.grad is not None:
                raise RuntimeError("BN input consumed twice")
            z.set_grad(dz)
        if want_res:
            if res.grad is not None:
                raise RuntimeError("residual gradient slot already filled")
            res.set_grad(gm)

    def _bn_grad_views(self, layer):
        """[E,C] views for d gamma / d beta of ``layer`` (+ an optional store step; hook for channel-padded BatchNorms)."""
        return (self._grad_slot("gamma", layer).view(self.E, layer.C), self._grad_slot("beta", layer).view(self.E, layer.C),
                None)

    def _stem_tail(self, z2, stats):
        """relu(bn_c2(z2)) -> relu(bn1(.)) -> maxpool(3,2,1) without materialising the two intermediates
        (csrc/stem_tail.hip); backward re-derives them from z2 in three streaming passes."""
        E, C_ = self.E, self.bn_c2.C
        n, h, w, _ = z2.t.shape
        rpe = self.B * h * w
        sc2, sh2, mu2, is2 = self._bn_coeffs(self.bn_c2, rpe, stats, stats.shape[0] // E if stats is not None else 0, z2)
        nparts = self._nparts(rpe)
        part = part_x = None
        if self.training:
            part = torch.empty(E, nparts, 2, C_, dtype=F32, device=self.dev)
            shc = torch.empty(E, C_, dtype=F32, device=self.dev)
            if self.taping:      # channel moments of z2 for the pooled-pass backward (stem_tail.hip)
                part_x = torch.empty(E, nparts, 3, C_, dtype=F32, device=self.dev)
            ops.stem_tail_stats(z2.t, sc2, sh2, mu2, part, nparts, E, self.B, shiftc=shc, part_x=part_x)
            if part_x is not None:
                part_x, npx = self._fold_parts(part_x, nparts, 3 * C_)
        sc1, sh1, mu1, is1 = self._bn_coeffs(self.bn1, rpe, part, nparts, shiftc=shc if self.training else None)
        ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        y = Var(self._new(n, ho, wo, C_))
        am = torch.empty(n, ho, wo, C_, dtype=torch.uint8, device=self.dev)
        ops.stem_tail_pool(z2.t, y.t, am, sc2, sh2, sc1, sh1, mu2, mu1, self.B)
        if self.debug_acts is not None:
            self.debug_acts["stem_tail"] = (am, y.t)      # winning tap | 0x80 if the winner's a2 > 0; a3 > 0 <=> y > 0
        y.needs_grad = z2.needs_grad or self.bn_c2.trainable or self.bn1.trainable
        if self.taping and y.needs_grad:
            train = self.training

            def bwd():
                dp = y.grad
                if dp is None:
                    return
                c11, c21, c12, c22 = (torch.empty(E, C_, dtype=F32, device=self.dev) for _ in range(4))
                consts = [sc2, sh2, sc1, sh1, mu1, is1, mu2, is2, c11, c21, c12, c22]
                p1 = torch.empty(E, nparts, 2, C_, dtype=F32, device=self.dev)
                if train and part_x is not None and self.pooled_stem_bwd:
                    # both reductions from ONE pass over the pooled tensors + the forward moments of z2
                    np4 = self._nparts(self.B * ho * wo)
                    p4 = torch.empty(E, np4, 4, C_, dtype=F32, device=self.dev)
                    ops.stem_tail_pooled(y.t, dp, am, consts, p4, np4, E)
                    p4, np4 = self._fold_parts(p4, np4, 4 * C_)
                    o1, o2 = (torch.empty(E, 1, 2, C_, dtype=F32, device=self.dev) for _ in range(2))
                    ops.stem_tail_combine(p4, np4, part_x, npx, consts, rpe, o1, o2, E, C_)
                    ops.bn_bwd_finalize(o1, 1, rpe, self._grad_slot("gamma", self.bn1).view(E, C_),
                                        self._grad_slot("beta", self.bn1).view(E, C_), c11, c21, E, C_)
                    ops.bn_bwd_finalize(o2, 1, rpe, self._grad_slot("gamma", self.bn_c2).view(E, C_),
                                        self._grad_slot("beta", self.bn_c2).view(E, C_), c12, c22, E, C_)
                else:
                    ops.stem_tail_bwd(1, z2.t, dp, am, None, consts, p1, nparts, E, self.B)
                    pf, nf = self._fold_parts(p1, nparts, 2 * C_)
                    ops.bn_bwd_finalize(pf, nf, rpe, self._grad_slot("gamma", self.bn1).view(E, C_),
                                        self._grad_slot("beta", self.bn1).view(E, C_), c11, c21, E, C_)
                    if not train:
                        c11.zero_()
                        c21.zero_()
                    ops.stem_tail_bwd(2, z2.t, dp, am, None, consts, p1, nparts, E, self.B)
                    pf, nf = self._fold_parts(p1, nparts, 2 * C_)
                    ops.bn_bwd_finalize(pf, nf, rpe, self._grad_slot("gamma", self.bn_c2).view(E, C_),
                                        self._grad_slot("beta", self.bn_c2).view(E, C_), c12, c22, E, C_)
                    if not train:
                        c12.zero_()
                        c22.zero_()
                if z2.needs_grad:
                    dz = torch.empty_like(z2.t)
                    ops.stem_tail_bwd(3, z2.t, dp, am, dz, consts, p1, nparts, E, self.B)
                    z2.set_grad(dz)
            self.tape.append(bwd)
        return y

    def _maxpool(self, x):
        n, h, w, c = x.t.shape
        ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        y = Var(self._new(n, ho, wo, c))
        am = torch.empty(n, ho, wo, c, dtype=torch.uint8, device=self.dev)
        ops.maxpool_fwd(x.t, y.t, am)
        if self.debug_acts is not None:
            self.debug_acts["maxpool"] = (am, 0, c)
        y.needs_grad = x.needs_grad
        if self.taping and y.needs_grad:
            def bwd():
                if y.grad is None:
                    return
                dx = torch.empty_like(x.t)
                ops.maxpool_bwd(y.grad, am, dx)
                x.set_grad(dx)
            self.tape.append(bwd)
        return y

    def _gap_parts(self, hw):
        # partial rows per image: >= 1024 pixels each, and enough workgroups (images x parts) to stream at HBM rate when
        # the batch is small (stage-1 U-Net, B = 10: 16 parts left 160 workgroups on 256 CUs)
        want = max(16, -(-2048 // max(1, self.N)))
        return max(1, min(want, hw // 1024))

    def _eca(self, x, layer, shared, tape=True):
        """y[n] = x[n or n % B] * sigmoid(conv1d(GAP(x)))  (EfficientBlock, basics.py:69-76)."""
        nx, h, w, c = x.t.shape
        hw = h * w
        nparts = self._gap_parts(hw)
        part = torch.empty(nx, nparts, c, dtype=F32, device=self.dev)
        ops.gap_partial(x.t, None, part, nparts)
        gate = torch.empty(self.N, c, dtype=F32, device=self.dev)
        gapmean = torch.empty(self.N, c, dtype=F32, device=self.dev)
        ops.eca_gate(part, nparts, hw, self._tab("eca", layer), layer.k, gate, gapmean, self.N, self.B,
                     self.B if shared else 0, c, layer.creal)
        y = Var(self._new(self.N, h, w, c))
        ops.eca_scale(x.t, gate, y.t, self.B if shared else 0)
        y.needs_grad = x.needs_grad or layer.trainable
        if not tape:
            return y, gate, gapmean
        if self.taping and y.needs_grad:
            def bwd():
                dy = y.grad
                if dy is None:
                    return
                dot = torch.empty(self.N, nparts, c, dtype=F32, device=self.dev)
                ops.gap_partial(dy, x.t, dot, nparts, self.B if shared else 0)
                dgap = torch.empty(self.N, c, dtype=F32, device=self.dev)
                ops.eca_bwd_small(dot, nparts, gate, gapmean, self._tab("eca", layer), layer.k, dgap,
                                  self._grad_slot("eca", layer).view(self.E, layer.k), self.N, self.B, c, layer.creal)
                if x.needs_grad:
                    dx = torch.empty_like(x.t)
                    ops.eca_bwd_apply(dy, gate, dgap, dx)
                    x.set_grad(dx)
            self.tape.append(bwd)
        return y

    def _eca_conv_folded(self, x, ecal, layer):
        """conv(x * sigmoid(conv1d(GAP(x)))) with the gate folded into per-IMAGE weight packs (the conv runs with one
        "expert" per image), so neither the gated activation nor its gradient is ever written:
          forward : gap -> gate [N,C] -> pack W*g[n] -> conv(x, W_n)
          backward: per-image filter gradients G[n] = dy (x) x  ->  dW = sum_n g[n] G[n],  ds = sum W G[n]  (eca_stem_fold)
                    -> ECA weight gradient and dgap  ->  dx = conv^T(dy, W_n) + dgap/HW  (the per-image bias of that conv)"""
        nx, h, w, c = x.t.shape
        hw = h * w
        N, B_, E = self.N, self.B, self.E
        if getattr(x, "gap_part", None) is not None:          # left by the BatchNorm pass that wrote x (_bn, want_gap)
            part, nparts = x.gap_part
        else:
            nparts = self._gap_parts(hw)
            part = torch.empty(nx, nparts, c, dtype=F32, device=self.dev)
            ops.gap_partial(x.t, None, part, nparts)
        gate = torch.empty(N, c, dtype=F32, device=self.dev)
        gapmean = torch.empty(N, c, dtype=F32, device=self.dev)
        ops.eca_gate(part, nparts, hw, self._tab("eca", ecal), ecal.k, gate, gapmean, N, B_, 0, c, ecal.creal)
        wf = torch.empty(N, layer.coutp, layer.taps, layer.cinp, dtype=self.dtype, device=self.dev)
        wd = torch.empty(N, layer.dg_rows, layer.taps, layer.dg_red, dtype=self.dtype, device=self.dev) if x.needs_grad else None
        ops.pack_conv_weights_gated(self._tab("w", layer), gate, wf, wd, N, B_, layer.cout, layer.cin, layer.ks, layer.coutp,
                                    layer.cinp, layer.dg_rows, layer.dg_red, self.dtype)
        Ho = ops.conv_out_size(h, layer.ks, layer.stride, layer.pad)
        Wo = ops.conv_out_size(w, layer.ks, layer.stride, layer.pad)
        o = Var(self._new(N, Ho, Wo, layer.cout_st), layer.cout_st, 0)
        stats = None
        if self.training and self.fuse_conv_stats and self.dtype == torch.bfloat16:
            rows = ops.conv2d_stat_rows(N, h, w, Ho, Wo, layer.cinp, layer.cout_st, layer.coutp, 1, layer.ks, layer.stride,
                                        layer.pad, self.dtype, in_ld=x.t.shape[-1], out_ld=o.t.shape[-1])
            stats = torch.empty(rows, 2, layer.coutp, dtype=F32, device=self.dev)
        flop = 2.0 * N * Ho * Wo * layer.cout * layer.cin * layer.taps
        ops.set_meta(flop=flop, name=layer.name)
        ops.conv2d(x.t, wf, o.t, cin=layer.cinp, cout=layer.cout_st, coutp=layer.coutp, ipe=1, ks=layer.ks,
                   stride=layer.stride, pad=layer.pad, stats=stats)
        o.needs_grad = x.needs_grad or layer.trainable or ecal.trainable
        if self.taping and o.needs_grad:
            def bwd():
                dy = o.grad
                if dy is None:
                    return
                ckw = 64 if self.dtype == torch.bfloat16 else 32
                cpw = (layer.cinp + ckw - 1) // ckw * ckw
                cow = (layer.cout_st + ckw - 1) // ckw * ckw
                G = self._wgrad_ws(N * layer.taps * cow * cpw, main=True)
                ops.set_meta(flop=flop, name=layer.name)
                ops.conv2d_wgrad(x.t, dy, G, cin=layer.cinp, cout=layer.cout_st, cinp=cpw, coutp=cow, ipe=B_, ks=layer.ks,
                                 stride=layer.stride, pad=layer.pad, per_image=True)
                ds = torch.empty(N, c, dtype=F32, device=self.dev)
                dw = self._grad_slot("w", layer) if layer.trainable else torch.empty(
                    E * layer.cout * layer.cin * layer.taps, dtype=F32, device=self.dev)
                ops.eca_stem_fold(G, gate, self._tab("w", layer), dw, ds, N, B_, layer.cout, layer.cin, layer.ks, cow, cpw)
                dgap = torch.empty(N, c, dtype=F32, device=self.dev) if x.needs_grad else None
                if ecal.trainable or x.needs_grad:
                    dwe = self._grad_slot("eca", ecal).view(E, ecal.k) if ecal.trainable else torch.empty(
                        E, ecal.k, dtype=F32, device=self.dev)
                    ops.eca_bwd_small(ds, 1, gate, gapmean, self._tab("eca", ecal), ecal.k, dgap, dwe, N, B_, c, ecal.creal,
                                      dgap_scale=1.0 / hw)
                if x.needs_grad:
                    if x.grad is not None or x.act != hip.ACT_NONE:
                        raise RuntimeError("gate-folded conv: its input must have this conv as the only consumer")
                    g = torch.empty_like(x.t)
                    kw = dict(cin=layer.dg_red, cout=layer.cinp, coutp=layer.dg_rows, ipe=1, ks=layer.ks, stride=1,
                              pad=layer.ks - 1 - layer.pad, dilate=False, in_coff=0, out_coff=0)
                    if not self._dgrad_with_bn_reduce(x, dy, wd, g, kw, flop, layer.name, bias=dgap):
                        ops.set_meta(flop=flop, name=layer.name + ":dgrad")
                        ops.conv2d(dy, wd, g, bias=dgap, **kw)
                        x.set_grad(g)
            self.tape.append(bwd)
        return o, stats

    def _gap_to(self, x, feat, coff):
        n, h, w, c = x.t.shape
        hw = h * w
        nparts = self._gap_parts(hw)
        part = torch.empty(n, nparts, c, dtype=F32, device=self.dev)
        ops.gap_partial(x.t, None, part, nparts)
        ops.gap_finish(part, feat.t, n, c, nparts, hw, feat.t.shape[-1], coff)
        if x.needs_grad:
            feat.needs_grad = True
        if self.taping and x.needs_grad:
            def bwd():
                g = feat.grad
                if g is None:
                    return
                dx = torch.empty_like(x.t)
                ops.gap_bwd(g, dx, g.shape[-1], coff)
                x.set_grad(dx)
            self.tape.append(bwd)

    def _act(self, x, act, drop_p):
        """y = dropout(act(x)) as its own pass (behind a BatchNorm1d, basics.py:34-39); backward from the saved output."""
        y = Var(torch.empty_like(x.t))
        seed = (next(self._seed_counter) * 0x9E3779B1 + self.base_seed) & 0xFFFFFFFFFFFF if drop_p > 0 else 0
        ops.act_fwd(x.t, y.t, act, drop_p, seed)
        y.needs_grad = x.needs_grad
        if self.taping and y.needs_grad:
            def bwd():
                if y.grad is None:
                    return
                dx = torch.empty_like(x.t)
                ops.act_bwd(y.grad, y.t, dx, act, drop_p)
                x.set_grad(dx)
            self.tape.append(bwd)
        return y

    def _mlp(self, x, spec, out=None, out_coff=0, in_shared=False):
        layers = spec["layers"]
        drop = spec["dropout"] if self.training else 0.0
        v = x
        if spec.get("bns") is not None:
            # make_mlp(bn=True): Linear (no bias) -> BatchNorm1d over the expert's batch rows -> act -> Dropout per hidden
            # layer, bare Linear last (basics.py:30-42; docs/experiments.md, conf/stage_3.yaml)
            for i, layer in enumerate(layers):
                last = i == len(layers) - 1
                if last:
                    act = spec["act"] if spec["l_act"] else hip.ACT_NONE
                    return self._conv(v, layer, act=act, out=out, out_coff=out_coff, in_shared=in_shared and i == 0)
                z = self._conv(v, layer, bias=False, in_shared=in_shared and i == 0)
                v = self._act(self._bn(z, spec["bns"][i], relu=False), spec["act"], drop)
            return v
        for i, layer in enumerate(layers):
            last = i == len(layers) - 1
            act = spec["act"] if (not last or spec["l_act"]) else hip.ACT_NONE
            dp = drop if not last else 0.0        # basics.py:33-39: dropout follows hidden activations only
            v = self._conv(v, layer, act=act, drop_p=dp, out=out if last else None, out_coff=out_coff if last else 0,
                           in_shared=in_shared and i == 0)
        return v

    # ------------------------------------------------------------------ gradient arena
    def _wgrad_ws(self, numel, main=False):
        """f32 scratch of the weight-gradient kernels.  Two buffers: the per-layer wgrads may run on the side stream
        (overlap_wgrad) while the per-image filter-gradient folds of the stem stay on the main stream."""
        key = "_ws_main" if main else "_ws"
        buf = self.__dict__.get(key)
        if buf is None or buf.numel() < numel or buf.device != self.dev:
            with torch.cuda.stream(torch.cuda.default_stream(self.dev)):      # owned by no transient stream's pool
                buf = torch.empty(numel, dtype=F32, device=self.dev)
            self.__dict__[key] = buf
        return buf[:numel]

    @staticmethod
    def _key(kind, layer):
        return (kind, id(layer) if not isinstance(layer, tuple) else (id(layer[0]), layer[1]))

    def _layout_arena(self):
        """Gradient arena in BACKWARD order (heads first, stem last): contiguous all-reduce buckets."""
        self._slots, self._order = {}, []
        off = 0
        for kind, layer, plist in reversed(self.params):
            n = sum(p.numel() for p in plist)
            key = self._key(kind, layer)
            self._slots[key] = (off, n, plist)
            self._order.append(key)
            off += n
        self._arena_used = off
        # padded so that every bucket boundary (multiples of 256 elements) divides by any world size up to 256: the
        # reduce-scatter / all-gather buckets of pmoe_amd.parallel need length % world == 0
        self._arena_numel = (off + 255) // 256 * 256

    def _bucket_cuts(self, n_buckets):
        """Bucket END offsets of the data-parallel exchange, by backward TIME rather than bytes.  The arena is in backward
        order (heads, layer4, ..., layer1, stem): the bytes sit at its head (layer4 holds 60 % of an expert's parameters)
        while the time sits at its tail (stem + layer1: ~45 % of backward for < 2 % of the parameters).  The last bucket
        completes only when backward ends and is therefore never hidden, so it holds just that tail; the buckets before it
        split the rest evenly by bytes."""
        tail = None
        for key in self._order:
            layer = next((l for k, l, _ in self.params if self._key(k, l) == key), None)
            name = getattr(layer[0] if isinstance(layer, tuple) else layer, "name", "")
            if name.startswith("layer1.") or name.startswith("stem.") or name in ("bn1", "eca1", "eca2"):
                tail = self._slots[key][0]
                break
        n = self._arena_numel
        if tail is None or tail <= 0 or n_buckets < 2:
            b = (n + n_buckets - 1) // n_buckets
            return [min(n, (i + 1) * ((b + 255) // 256 * 256)) for i in range(n_buckets)]
        tail = tail // 256 * 256
        b = (tail + n_buckets - 2) // (n_buckets - 1)
        b = (b + 255) // 256 * 256
        return [min(tail, (i + 1) * b) for i in range(n_buckets - 1)] + [n]

    def _grad_slot(self, kind, layer):
        key = self._key(kind, layer)
        off, n, _ = self._slots[key]
        self._filled.add(key)
        return self._arena[off:off + n]

    def _final_prefix(self):
        """Arena offset below which every slot is either written or belongs to frozen parameters."""
        while self._cursor < len(self._order):
            key = self._order[self._cursor]
            _, _, plist = self._slots[key]
            if key in self._filled or not any(p.requires_grad for p in plist):
                self._cursor += 1
            else:
                break
        if self._cursor == len(self._order):
            return self._arena_numel
        return self._slots[self._order[self._cursor]][0]

    # ------------------------------------------------------------------ network
    def _begin(self, images, training, taping, dtype, base_seed):
        if images.dim() != 5:
            raise ValueError(f"images: expected [B,T,C,H,W], got {tuple(images.shape)}")
        if not images.is_cuda:
            raise RuntimeError("pmoe_amd: inputs must be on the MI355X (cuda) device; there is no CPU path")
        hip.load()
        self.dev, self.dtype = images.device, dtype
        self.training, self.taping = training, taping
        self.base_seed = int(base_seed)
        Bsz = images.shape[0]
        self.B, self.N = Bsz, Bsz * self.E
        self._ensure_built(self.dev, dtype)
        self._refresh_tables(self.dev)
        self._pack_all()
        self.tape, self._bn_touched = [], []
        self._seed_counter = itertools.count(1)       # dropout masks are a function of (base_seed, layer order)

        return Bsz

    def forward(self, images, speed, command, training, taping, dtype, base_seed=0):
        """Returns probs [B,K], mean [B,K,2], std [B,K,2], speeds [B,K,1] ([B,1] for the shared trunk), f32, and
        the tape."""
        Bsz = self._begin(images, training, taping, dtype, base_seed)
        x0 = self._image_input(images)
        spd, cmd = self._measurement_inputs(speed, command)

        feat = Var(self._new(self.N, 1, 1, 1536))
        self._backbone_fwd(x0, feat)
        # ---- measurement encoders write straight into their slots of the 1536-d feature (moe.py:88-95)
        self._mlp(spd, self.speed_enc, out=feat, out_coff=512, in_shared=True)
        self._mlp(cmd, self.cmd_enc, out=feat, out_coff=1024, in_shared=True)
        return self._heads_fwd(feat, Bsz, training)

    def _backbone_fwd(self, x0, feat):
        """x0 [B,H,W,Cp] (shared by all experts) -> feat[:, 0:512]: ResNet18 body with the ECA stem
        (backbone.py:63-70, basics.py:79-134)."""
        H, W = x0.t.shape[1], x0.t.shape[2]
        hw_ok = H * W >= 256                                 # per-image filter gradients need one tile <= one image
        fold2 = self.fold_eca_gate and hw_ok and self.conv2.cin % 64 == 0
        if self.fold_stem_input and self.taping and hw_ok and (self.conv1.trainable or self.eca1.trainable):
            # backward of (ECA gate -> conv1) from per-image filter gradients: no data-gradient conv (heads.hip)
            x0s, gate, gapmean = self._eca(x0, self.eca1, shared=True, tape=False)
            z1, st = self._conv_stats(x0s, self.conv1, tape=False)
            z1.needs_grad = True
            # round 4: bn_c1's backward apply happens inside conv1's per-image filter gradient (dz1, 2.15 GB at the headline shape,
            # is never written): possible because that launch is the only consumer of dz1 (PMOE_STEM_BN_FUSE=0: A/B switch)
            z1.bn_defer = (self.stem_bn_fuse and self.training and self.bn_reduce_in_dgrad and self.dtype == torch.bfloat16
                           and fold2 and ops.conv2d_wgrad(x0.t, z1.t, None, cin=self.conv1.cinp, cout=self.conv1.cout_st,
                                                          cinp=64, coutp=64, ipe=self.B, ks=3, stride=1, pad=self.conv1.pad,
                                                          x_shared=True, per_image=True, bn_fuse=(z1.t, None, None, None),
                                                          plan_only=True) == 7209)
            self.tape.append(lambda: self._stem_in_bwd(x0, z1, gate, gapmean))
            a1 = self._bn(z1, self.bn_c1, relu=True, stats=st, want_gap=fold2)
        else:
            x0s = self._eca(x0, self.eca1, shared=True)
            a1 = self._conv_bn(x0s, self.conv1, self.bn_c1, relu=True)
        if fold2:
            z2, st = self._eca_conv_folded(a1, self.eca2, self.conv2)
        else:
            a1s = self._eca(a1, self.eca2, shared=False)
            z2, st = self._conv_stats(a1s, self.conv2)
        if self.fuse_stem_tail:
            o = self._stem_tail(z2, st)                        # BN+ReLU, bn1+ReLU, maxpool in one fused chain
        else:
            a2 = self._bn(z2, self.bn_c2, relu=True, stats=st)
            a3 = self._bn(a2, self.bn1, relu=True)             # torchvision bn1 + relu stay after the stem
            o = self._maxpool(a3)
        for bi, blk in enumerate(self.blocks):
            idn = o
            if blk["down"] is not None:
                # before conv1 on the tape, so that in backward the 1x1 stride-2 data gradient runs LAST and is added in
                # place at the even pixels of conv1's (dense) data gradient instead of writing a mostly-zero tensor first
                idn = self._conv_bn(o, blk["down"][0], blk["down"][1], relu=False)
            # fp8 policy: a BatchNorm pass whose output feeds an fp8 convolution also leaves it as e4m3 bytes
            nxt = self.blocks[bi + 1]["conv1"] if bi + 1 < len(self.blocks) else None
            aA = self._conv_bn(o, blk["conv1"], blk["bn1"], relu=True, want_f8=blk["conv2"].fp8)
            o = self._conv_bn(aA, blk["conv2"], blk["bn2"], relu=True, res=idn, want_f8=nxt is not None and nxt.fp8)
        self._gap_to(o, feat, 0)

    def _conv_bn(self, x, conv, bn, relu, res=None, out=None, want_f8=False, pool_to=None):
        """[relu](bn(conv(x)) [+ res]).  Training / taped: conv (+ fused statistics) then the BatchNorm passes.
        Inference (eval mode, nothing taped): the BatchNorm is FOLDED into the conv -- weights scaled per output channel,
        beta - mean*scale as the bias, residual add and ReLU in the conv epilogue: no pass over the activation at all."""
        if self.training or self.taping or not self.fold_bn_eval or conv.w_f8 is not None:      # (the fp8 policy is not folded)
            z, st = self._conv_stats(x, conv)
            return self._bn(z, bn, relu=relu, res=res, stats=st, out=out, want_f8=want_f8, pool_to=pool_to)
        E = self.E
        key = (self.dtype, str(self.dev)) + tuple(p._version for p in conv.weights) + tuple(
            v for m in bn.mods for v in (m.weight._version, m.bias._version, m.running_mean._version, m.running_var._version))
        cache = conv.__dict__.get("_bn_fold")
        if cache is None or cache[0] != key:
            scale, shift, mean, _ = self._bn_coeffs(bn, 1)
            wf = torch.empty(E, conv.coutp, conv.taps, conv.cinp, dtype=self.dtype, device=self.dev)
            bf = torch.empty(E, conv.coutp, dtype=F32, device=self.dev)
            ops.pack_conv_weights_scaled(self._tab("w", conv), scale, shift, mean, wf, bf, E, conv.cout, conv.cin, conv.ks,
                                         conv.coutp, conv.cinp, self.dtype)
            cache = conv.__dict__["_bn_fold"] = (key, wf, bf)
        _, wf, bf = cache
        H, W = x.t.shape[1], x.t.shape[2]
        Ho = ops.conv_out_size(H, conv.ks, conv.stride, conv.pad)
        Wo = ops.conv_out_size(W, conv.ks, conv.stride, conv.pad)
        o = Var(self._new(self.N, Ho, Wo, conv.cout_st), conv.cout_st, 0) if out is None else out.window(0, conv.cout_st)
        ops.set_meta(flop=2.0 * self.N * Ho * Wo * conv.cout * conv.cin * conv.taps, name=conv.name + "+bn")
        ops.conv2d(x.t, wf, o.t, cin=conv.cinp, cout=conv.cout_st, coutp=conv.coutp, ipe=self.B, ks=conv.ks,
                   stride=conv.stride, pad=conv.pad, in_shared=(x.t.shape[0] != self.N), in_coff=x.coff, out_coff=o.coff,
                   bias=bf,
                   act=hip.ACT_RELU if relu else hip.ACT_NONE, res=res.t if res is not None else None,
                   res_coff=res.coff if res is not None else 0, res_mode=hip.RES_ADD)
        return o

    def _heads_fwd(self, feat, Bsz, training):
        E = self.E
        sp = self._mlp(feat, self.speed_pred)
        af = self._mlp(feat, self.action_feat)
        if self.alt:
            head = self._conv(af, self.head)                     # [N,1,1,16]: mean(2) raw-std(2)
            al = self._mlp(feat, self.alpha_mlp)                 # [N,1,1,16]: col 0 = alpha
            head5 = self._merge_alt_head(head, al)
        else:
            head5 = self._conv(af, self.head)                    # cols 0..3 action_pred, col 4 alpha
        K = self.K
        probs = torch.empty(Bsz, K, dtype=F32, device=self.dev)
        mean = torch.empty(Bsz, K, 2, dtype=F32, device=self.dev)
        std = torch.empty(Bsz, K, 2, dtype=F32, device=self.dev)
        speeds = torch.empty((Bsz, 1) if self.shared else (Bsz, K, 1), dtype=F32, device=self.dev)
        # BaseExpert applies ReLU to alpha (moe.py:97); BaseExpertAlt (moe.py:126) and the shared head (moe.py:226) do not
        ops.gate_mixture_fwd(head5.t.view(self.N, -1), sp.t.view(self.N, 16), probs, mean, std, speeds, Bsz, K,
                             self._alpha_mode(), self.shared)
        self._bump_batch_counters()
        state = dict(tape=self.tape, tail=(head5, sp, probs), B=self.B, N=self.N, dev=self.dev, dtype=self.dtype)
        self.tape = None
        return probs, mean, std, speeds, state

    def _alpha_mode(self):
        # bit 0: BaseExpert applies ReLU to alpha (moe.py:97); BaseExpertAlt (moe.py:126) and the shared head (moe.py:226)
        # do not.  bit 1: lone expert, no softmax (pmoe_gate_mixture_fwd)
        return int(not self.alt and not self.shared) | (2 if self.raw_alpha else 0)

    def _measurement_inputs(self, speed, command):
        Bsz = self.B
        if speed.shape != (Bsz, self.speed_enc["layers"][0].cin) or command.shape != (Bsz, self.cmd_enc["layers"][0].cin):
            raise ValueError("speed / command shapes do not match the encoders")
        spd = Var(self._new(Bsz, 1, 1, 16))
        ops.pad_rows(speed.contiguous().float(), spd.t.view(Bsz, 16))
        cmd = Var(self._new(Bsz, 1, 1, 16))
        ops.pad_rows(command.contiguous().float(), cmd.t.view(Bsz, 16))
        return spd, cmd

    def _bump_batch_counters(self):
        """num_batches_tracked += (train-mode passes through that BatchNorm during this forward)."""
        if not self._bn_touched:
            return
        count, layers = {}, {}
        for l in self._bn_touched:
            count[id(l)] = count.get(id(l), 0) + 1
            layers[id(l)] = l
        tensors = [m.num_batches_tracked for k, l in layers.items() for m in l.mods]
        steps = [count[k] for k, l in layers.items() for _ in l.mods]
        torch._foreach_add_(tensors, steps)
        # bn_finalize updated the running statistics through raw pointers: bump their version counters so that every
        # cache keyed on them (the eval-mode BatchNorm fold of _conv_bn) sees the change, as nn.BatchNorm2d's own
        # in-place update would have made visible
        torch.autograd.graph.increment_version([b for l in layers.values() for m in l.mods
                                                for b in (m.running_mean, m.running_var)])

    def _image_input(self, images):
        """[B,T,C,H,W] f32 -> NHWC [B,H,W,r16(T*C)] in the compute dtype (frames concatenated along channels, moe.py:90-92)."""
        Bsz, (H, W) = images.shape[0], images.shape[-2:]
        cin = images.shape[1] * images.shape[2]
        if cin != self.conv1.cin:
            raise ValueError(f"images carry {cin} channels (T*C), the backbone stem expects {self.conv1.cin}")
        x0 = Var(self._new(Bsz, H, W, r16(cin)))
        ops.nchw_to_nhwc(images.reshape(Bsz, cin, H, W).contiguous().float(), x0.t)
        return x0

    def _conv_stats(self, x, layer, tape=True, in_bn=None):
        # conv-epilogue statistics are plain sums (no sample to centre on before the conv has run): fine under
        # bf16 storage noise, not for the exact-f32 parity mode, which takes the centred colstats pass instead
        if self.training and self.fuse_conv_stats and self.dtype == torch.bfloat16:
            return self._conv(x, layer, bias=False, want_stats=True, tape=tape, in_bn=in_bn)
        return self._conv(x, layer, bias=False, tape=tape, in_bn=in_bn), None

    def _stem_in_bwd(self, x0, z1, gate, gapmean):
        dy = z1.grad
        if dy is None:
            return
        E, layer, ecal = self.E, self.conv1, self.eca1
        ckw = 64 if self.dtype == torch.bfloat16 else 32
        cpw = (layer.cinp + ckw - 1) // ckw * ckw
        cow = (layer.cout_st + ckw - 1) // ckw * ckw
        G = self._wgrad_ws(self.N * layer.taps * cow * cpw, main=True)
        ops.set_meta(flop=2.0 * self.N * dy.shape[1] * dy.shape[2] * layer.cout * layer.cin * layer.taps, name=layer.name,
                     bytes=(2 if z1.bn_fused is not None else 1) * dy.numel() * dy.element_size())
        fuse = None
        if z1.bn_fused is not None:                      # dy is g (masked gradient of the BatchNorm output): dz1 on load
            g_, coef, c1, c2 = z1.bn_fused
            fuse = (z1.t, coef, c1, c2)
        ops.conv2d_wgrad(x0.t, dy, G, cin=layer.cinp, cout=layer.cout_st, cinp=cpw, coutp=cow, ipe=self.B, ks=layer.ks,
                         stride=1, pad=layer.pad, x_shared=True, per_image=True, bn_fuse=fuse)
        ds = torch.empty(self.N, gate.shape[-1], dtype=F32, device=self.dev)
        dw = self._grad_slot("w", layer) if layer.trainable else torch.empty(E * layer.cout * layer.cin * layer.taps,
                                                                             dtype=F32, device=self.dev)
        ops.eca_stem_fold(G, gate, self._tab("w", layer), dw, ds, self.N, self.B, layer.cout, layer.cin, layer.ks, cow, cpw)
        if ecal.trainable:
            # ds plays the role of the (single) partial row of sum_hw dy*x in the generic ECA backward
            ops.eca_bwd_small(ds, 1, gate, gapmean, self._tab("eca", ecal), ecal.k, None,
                              self._grad_slot("eca", ecal).view(E, ecal.k), self.N, self.B, gate.shape[-1], ecal.creal)

    def _merge_alt_head(self, head, al):
        """moe_alt: alpha comes from its own MLP; place it in column 4 of the head rows (device copy)."""
        merged = Var(head.t.clone())
        merged.t.view(self.N, 16)[:, 4] = al.t.view(self.N, 16)[:, 0]
        merged.needs_grad = head.needs_grad or al.needs_grad
        if self.taping and merged.needs_grad:
            def bwd():
                g = merged.grad
                if g is None:
                    return
                gh = g.clone()
                gh.view(self.N, 16)[:, 4] = 0
                head.set_grad(gh)
                ga = torch.zeros_like(al.t)
                ga.view(self.N, 16)[:, 0] = g.view(self.N, 16)[:, 4]
                al.set_grad(ga)
            self.tape.append(bwd)
        return merged

    def _tail_bwd(self, tail, dprobs, dmean, dstd, dspeeds):
        head5, sp, probs = tail
        dhead = torch.empty_like(head5.t)
        dspd = torch.empty_like(sp.t)

        def c(t):
            return t.contiguous().float() if t is not None else None
        ops.gate_mixture_bwd(head5.t.view(self.N, -1), probs, c(dprobs), c(dmean), c(dstd), c(dspeeds),
                             dhead.view(self.N, -1), dspd.view(self.N, 16), self.B, self.K,
                             self._alpha_mode(), self.shared)
        if dprobs is not None or dmean is not None or dstd is not None:
            head5.set_grad(dhead)
        if dspeeds is not None:
            sp.set_grad(dspd)

    def backward(self, tape_state, *douts):
        """Run the recorded tape in reverse; returns the flat gradient arena and per-parameter views."""
        tape = tape_state["tape"]
        self.B, self.N, self.dev, self.dtype = (tape_state[k] for k in ("B", "N", "dev", "dtype"))
        self._layout_arena()
        self._arena = torch.zeros(self._arena_numel, dtype=F32, device=self.dev)
        self._filled, self._cursor = set(), 0
        reducer = None
        if self.dp_group is not None or (dist.is_initialized() and self.dp_enabled):
            reducer = BucketedAllReduce(self.dp_group, self.dp_buckets, self.dp_always)
            reducer.begin(self._arena, self._bucket_cuts(self.dp_buckets))
        self._tail_bwd(tape_state["tail"], *douts)
        main = torch.cuda.current_stream()
        side = self._side_stream() if self.overlap_wgrad else None
        for fn in reversed(tape):
            fn()
            if reducer is not None:
                # every launch of the finished closures is enqueued (weight gradients possibly on the side stream):
                # buckets below the prefix can fly once BOTH streams have reached this point
                if side is not None:
                    side.wait_stream(main)
                    with torch.cuda.stream(side):
                        reducer.ready(self._final_prefix())
                else:
                    reducer.ready(self._final_prefix())
        if reducer is not None:
            if side is not None:
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    reducer.finish()
            else:
                reducer.finish()
        if side is not None:
            main.wait_stream(side)
            self._arena.record_stream(side)
        grads = {}
        for key, (off, n, plist) in self._slots.items():
            o = off
            for p in plist:
                grads[id(p)] = self._arena[o:o + p.numel()].view_as(p) if key in self._filled else None
                o += p.numel()
        return grads
