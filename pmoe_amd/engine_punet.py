"""Engine for ``PUNetExpert`` (``PMoE/model/moe.py:268-323``): frozen PU-Net -> ResNet18-ECA backbone -> tanh head.

Built from the same primitive launches as :class:`pmoe_amd.engine.ExpertGroupEngine` (group of one):

* the PU-Net (``model/punet.py:75-120``) is FORWARD ONLY -- ``freeze(self.punet)`` (moe.py:280) removes every one of
  its parameters from training, so nothing is taped for it and the 138-channel stem needs no data gradient
  (``eca1`` / ``conv1`` gradients come from the per-image filter-gradient fold, as in the MoE path);
* train-mode BatchNorm inside the frozen U-Nets still uses batch statistics and updates its running buffers
  (``model.train()`` in train_2.py:130 reaches them), exactly like the reference: the four past frames go through
  ``unet`` one after the other, each with its own statistics;
* ``ConvTranspose2d(k=2,s=2)`` is one 1x1 GEMM with 4*Cout rows + ``pmoe_pixel_shuffle2`` into the second half of
  the skip-concatenation buffer; ``torch.cat`` / ``view`` of 23-class masks are ``pmoe_copy_window`` launches.
"""
import os

import torch

from . import hip, ops
from .engine import ExpertGroupEngine, GroupedBN, GroupedConv, Var, r16, F32


class _UpConv(GroupedConv):
    """ConvTranspose2d(cin, cout, 2, 2) as a 1x1 layer with 4*cout rows (row (dy*2+dx)*cout + c)."""

    def __init__(self, eng, name, mod):
        super().__init__(eng, name, None, None, mod.in_channels, 4 * mod.out_channels, 1, 1, 0)
        self.mod, self.c_up = mod, mod.out_channels
        self.need_dgrad = False
        self.biases = [mod.bias]          # marks "has bias" for alloc(); packed from the derived tensor below
        self._derived_version = None

    @property
    def trainable(self):
        return self.mod.weight.requires_grad or self.mod.bias.requires_grad

    def store_grads(self, eng, ws, cow, cpw):
        """weight gradient of the 4*Cout-row 1x1 layer -> ConvTranspose2d layout [cin, cout, 2, 2] (a strided copy);
        the bias gradient is taken from the un-shuffled output gradient in PUNetEngine._up_bwd."""
        full = torch.empty(1, self.cout, self.cin, dtype=F32, device=eng.dev)
        ops.unpack_conv_wgrad(ws, full, 1, self.cout, self.cin, 1, cow, cpw)
        eng._grad_slot("wT", self).view(self.cin, self.c_up, 2, 2).copy_(
            full.view(2, 2, self.c_up, self.cin).permute(3, 2, 0, 1))

    def pack_derived(self, dev):
        m = self.mod
        ver = (m.weight._version, m.bias._version, m.weight.data_ptr())
        if ver == self._derived_version and self.w_fwd is not None:
            return
        # [cin, cout, 2, 2] -> [(dy, dx, cout), cin, 1, 1]: parameter re-layout of a frozen weight (host-side plumbing)
        w = m.weight.detach().permute(2, 3, 1, 0).reshape(self.cout, self.cin, 1, 1).contiguous()
        b = m.bias.detach().repeat(4).contiguous()
        ops.pack_conv_weights(hip.ptr_table([w], dev), self.w_fwd, self.w_dg, 1, self.cout, self.cin, 1, self.coutp,
                              self.cinp, self.dg_rows, self.dg_red, self.w_fwd.dtype)
        ops.pack_bias(hip.ptr_table([b], dev), self.bias_packed, 1, self.cout, self.coutp)
        self._derived_version = ver


class PUNetEngine(ExpertGroupEngine):
    _punet_trains = False

    def __init__(self, expert):
        self.return_inter = expert.return_inter
        super().__init__([expert], alt=False)

    # ------------------------------------------------------------------ structure
    def _collect_pre_backbone(self, ex):
        pu = ex[0].punet
        self.pu = pu
        self.up_layers = []
        self.shadow_bns = []
        self.unet = self._collect_unet("punet.unet", pu.unet)
        conv, bn, eca = self._mk["conv"], self._mk["bn"], self._mk["eca"]
        eb = pu.entry_block
        self.entry = dict(eca1=eca("punet.entry.eca1", [eb.layer1.eca1]),
                          conv1=conv("punet.entry.conv1", [eb.layer1.conv1[0]]),
                          bn1=bn("punet.entry.bn1", [eb.layer1.conv1[1]]),
                          eca2=eca("punet.entry.eca2", [eb.layer2.eca2]),
                          conv2=conv("punet.entry.conv2", [eb.layer2.conv2[0]]),
                          bn2=self._padded_bn("punet.entry.bn2", eb.layer2.conv2[1]))
        self.pred_unet = self._collect_unet("punet.pred_unet", pu.pred_unet)
        for layer in self._punet_convs():
            layer.need_dgrad = self._punet_trains        # stage 2 freezes the PU-Net: no data gradients at all
        for U in (self.unet, self.pred_unet):
            for up in U["up"]:
                up.need_dgrad = self._punet_trains

    def _punet_convs(self):
        out = []
        for U in (self.unet, self.pred_unet):
            for blk in U["dwn"] + U["up_forw"]:
                out += [blk["c1"], blk["c2"]]
            out.append(U["out"])
        return out + [self.entry["conv1"], self.entry["conv2"]]

    def _padded_bn(self, name, mod):
        """BatchNorm over C channels stored in r16(C)-wide rows (C = 3 after the entry block): the kernels index
        whole rows, so gamma/beta/running stats are mirrored in zero-padded device buffers around each forward."""
        layer = GroupedBN(name, [mod])
        if mod.num_features % 16 == 0:
            self.params.append(("gamma", layer, [mod.weight]))
            self.params.append(("beta", layer, [mod.bias]))
            return layer
        layer.C = r16(mod.num_features)
        layer.creal = mod.num_features
        layer.shadow = None
        self.shadow_bns.append(layer)
        self.params.append(("gamma_real", layer, [mod.weight]))     # kept in the flat parameter list (frozen: no slot use)
        self.params.append(("beta_real", layer, [mod.bias]))
        return layer

    def _bn_grad_views(self, layer):
        creal = getattr(layer, "creal", None)
        if creal is None:
            return super()._bn_grad_views(layer)
        dg, db = (torch.empty(1, layer.C, dtype=F32, device=self.dev) for _ in range(2))

        def store():
            self._grad_slot("gamma_real", layer).copy_(dg[0, :creal])
            self._grad_slot("beta_real", layer).copy_(db[0, :creal])
        return dg, db, store

    def _collect_unet(self, name, U):
        conv, bn = self._mk["conv"], self._mk["bn"]

        def block(nm, seq):
            return dict(c1=conv(f"{nm}.0", [seq[0]]), bn1=bn(f"{nm}.1", [seq[1]]),
                        c2=conv(f"{nm}.3", [seq[3]]), bn2=bn(f"{nm}.4", [seq[4]]))

        d = dict(mod=U, dwn=[block(f"{name}.dwn_{i}", getattr(U, f"dwn_{i}")) for i in range(1, 6)], up=[], up_forw=[])
        for i in range(1, 5):
            up = _UpConv(self, f"{name}.up_{i}", getattr(U, f"up_{i}"))
            self.params.append(("wT", up, [up.mod.weight]))
            self.params.append(("bT", up, [up.mod.bias]))
            self.up_layers.append(up)
            d["up"].append(up)
            d["up_forw"].append(block(f"{name}.up_forw_{i}", getattr(U, f"up_forw_{i}")))
        d["out"] = conv(f"{name}.out", [U.out])
        return d

    def _collect_backbone(self, bbs):
        if self.return_inter:          # punet_inter: the PU-Net bottleneck vector is the image feature (moe.py:282-284)
            self.blocks = []
            self.conv1 = self.eca1 = None
            return
        super()._collect_backbone(bbs)
        c1 = self.conv1
        if self.pad_stem_input and c1.ks == 3 and c1.stride == 1 and c1.coutp == 64 and c1.cin > 64 and c1.cinp % 64:
            # round 4: the 138-channel stem input (F predicted masks x 23 classes) stored in rows of 192 channels instead of 144, zero
            # filled -- whole 64-channel chunks, so the convolution (667 GFLOP at the C4 shape) runs on the persistent LDS-DMA kernel's
            # 64-output-channel tiles (plan 5067) instead of the generic register-staged kernel's 16-channel chunks (273 TFLOP/s)
            c1.cinp = (c1.cin + 63) // 64 * 64

    pad_stem_input = True

    def _x0_width(self, channels):
        """row width of the backbone's input tensor: the stem convolution's (possibly 64-padded) input-channel count"""
        return self.conv1.cinp if self.conv1 is not None and self.conv1.cin == channels else r16(channels)

    def _collect_heads(self, ex):
        conv, mlp = self._mk["conv"], self._mk["mlp"]
        e = ex[0]
        self.speed_pred = mlp("speed_pred", [e.speed_pred])
        # action_pred = Sequential(make_mlp(action_head), Linear(512, 2))  (moe.py:296-301)
        self.action_feat = mlp("action_pred.0", [e.action_pred[0]])
        self.head = conv("action_pred.1", [e.action_pred[1]])

    def _ensure_built(self, dev, dtype):
        # (its own key: the base class stores a longer tuple in _built_for -- compared against that one, this branch ran on EVERY
        #  call, the fresh shadow tensors changed the pointer table, and all 79 weight packs of a PUNetExpert were redone every step:
        #  1.2 ms of pack launches + the derived ConvTranspose2d packs per step until round 4)
        key = (str(dev), dtype)
        if self.__dict__.get("_punet_built_for") != key:
            for up in self.up_layers:
                up.alloc(dtype, dev)
                up._derived_version = None
            for l in self.shadow_bns:
                l.shadow = {k: torch.zeros(l.C, dtype=F32, device=dev) for k in ("gamma", "beta", "rm", "rv")}
            self._punet_built_for = key
        super()._ensure_built(dev, dtype)

    def _extra_tables(self):
        return [((kind, id(l)), [l.shadow[kind]]) for l in self.shadow_bns for kind in ("gamma", "beta", "rm", "rv")]

    def _pack_all(self):
        for up in self.up_layers:
            up.pack_derived(self.dev)
        super()._pack_all()

    # ------------------------------------------------------------------ shadows of padded BatchNorms
    def _shadows_in(self):
        for l in self.shadow_bns:
            m, c = l.mods[0], l.creal
            l.shadow["gamma"][:c].copy_(m.weight.detach())
            l.shadow["beta"][:c].copy_(m.bias.detach())
            l.shadow["rm"][:c].copy_(m.running_mean)
            l.shadow["rv"].fill_(1.0)
            l.shadow["rv"][:c].copy_(m.running_var)

    def _shadows_out(self):
        for l in self.shadow_bns:
            m, c = l.mods[0], l.creal
            if m.training:
                m.running_mean.copy_(l.shadow["rm"][:c])
                m.running_var.copy_(l.shadow["rv"][:c])

    # ------------------------------------------------------------------ PU-Net forward
    fuse_bn_pool = True            # round 4: the down blocks' last BatchNorm + ReLU pass also writes MaxPool2d(2, 2) of its output

    fuse_in_bn = True      # round 4: BatchNorm + ReLU between the two convolutions of a block applied ON LOAD by the second one

    def _conv3(self, x, blk, out=None, pool_to=None, defer=False):
        """blocks/unet.py:14-24: (conv -> BatchNorm -> ReLU) x 2.  Untaped train-mode forward (the frozen U-Nets inside a training
        step): the activation between the two convolutions has exactly one consumer and nothing is saved for a backward pass, so
        where the second convolution's kernel can evaluate relu(bn1(z1)) on its halo patch (PMOE_RES_INBN: the 64-channel blocks)
        the pass that would write it -- and the tensor -- do not exist."""
        c1, bn1, c2 = blk["c1"], blk["bn1"], blk["c2"]
        if (self.fuse_in_bn and not self.taping and self.training and self.dtype == torch.bfloat16 and self.fuse_conv_stats
                and self.debug_acts is None and c2.w_f8 is None and c1.cout_st == c2.cinp == bn1.C):
            n, h, w, _ = x.t.shape
            ok = blk.get("_inbn")
            if ok is None or ok[0] != (n, h, w):
                probe = torch.empty(4, self.E, bn1.C, dtype=torch.float32, device=self.dev)
                zz = torch.empty(n, h, w, c1.cout_st, dtype=self.dtype, device=self.dev)
                oo = torch.empty(n, h, w, c2.cout_st, dtype=self.dtype, device=self.dev)
                code = ops.conv2d(zz, c2.w_fwd, oo, cin=c2.cinp, cout=c2.cout_st, coutp=c2.coutp, ipe=self.B, ks=c2.ks,
                                  stride=c2.stride, pad=c2.pad, res_mode=hip.RES_INBN, bn_coef=probe, plan_only=True)
                ok = blk["_inbn"] = ((n, h, w), code == 1267)
            if ok[1]:
                z1, st1 = self._conv_stats(x, c1)
                self._bn_coeffs(bn1, self.B * h * w, st1, st1.shape[0] // self.E, z1)
                z2, st2 = self._conv_stats(z1, c2, in_bn=self._last_coef)
                return self._last_bn(z2, st2, blk["bn2"], out, pool_to, defer)
        a = self._conv_bn(x, c1, bn1, relu=True)
        if (defer and self.fuse_in_bn and not self.taping and self.training and self.dtype == torch.bfloat16 and self.fuse_conv_stats
                and self.debug_acts is None and c2.w_f8 is None):
            z2, st2 = self._conv_stats(a, c2)
            return self._last_bn(z2, st2, blk["bn2"], out, pool_to, defer)
        return self._conv_bn(a, c2, blk["bn2"], relu=True, out=out, pool_to=pool_to)

    def _last_bn(self, z2, st2, bn2, out, pool_to, defer):
        """the block's second BatchNorm + ReLU: written out (`_bn`), or -- `defer`, untaped: the caller promises ONE consumer that is a
        1x1 layer -- finalized only: the statistics and running buffers are updated, the activation stays pending on z2 and the
        consumer applies it on load (conv1x1_direct_kernel<MT, true>) or `_materialize` writes it after all."""
        if not defer or out is not None or pool_to is not None:
            return self._bn(z2, bn2, relu=True, stats=st2, out=out, pool_to=pool_to)
        n, h, w, _ = z2.t.shape
        rpe = self.B * h * w
        self._bn_coeffs(bn2, rpe, st2, st2.shape[0] // self.E, z2)
        z2.pending_bn = (self._last_coef, rpe)
        return z2

    def _materialize(self, v):
        """v with a pending BatchNorm + ReLU (see _last_bn) -> the activation tensor, written by the pass the fused consumer avoids"""
        if v.pending_bn is None:
            return v
        coef, rpe = v.pending_bn
        y = Var(torch.empty_like(v.t))
        ops.bn_apply(v.t, None, y.t, coef[2], coef[3], coef[0], rpe, self.E, v.t.shape[-1], True)
        return y

    def _conv1x1_after_bn(self, h, layer, out=None):
        """1x1 layer (+ bias) over h: where h carries a pending BatchNorm + ReLU and the direct kernel serves the shape, applied on load"""
        if h.pending_bn is not None:
            n, hh, ww, _ = h.t.shape
            o = torch.empty(n, hh, ww, layer.cout_st, dtype=self.dtype, device=self.dev)
            kw = dict(cin=layer.cinp, cout=layer.cout_st, coutp=layer.coutp, ipe=self.B, ks=1, stride=1, pad=0, bias=layer.bias_packed)
            if ops.conv2d(h.t, layer.w_fwd, o, res_mode=hip.RES_INBN, bn_coef=h.pending_bn[0], plan_only=True, **kw) in (1412, 1414):
                ops.set_meta(flop=2.0 * n * hh * ww * layer.cout * layer.cin, name=layer.name + "+bn")
                ops.conv2d(h.t, layer.w_fwd, o, res_mode=hip.RES_INBN, bn_coef=h.pending_bn[0], **kw)
                return Var(o, layer.cout_st, 0)
            h = self._materialize(h)
        return self._conv(h, layer, bias=True)

    def _maxpool2(self, x, cat=None, fused=None):
        """``fused``: the pooled tensor was already written by the pass that produced x (_bn, pool_to): only the tape entry is added."""
        n, h, w, _ = x.t.shape
        y = Var(fused if fused is not None else self._new(n, h // 2, w // 2, x.c))
        if fused is None:
            ops.maxpool2_fwd(x.t, y.t, c=x.c, x_coff=x.coff)      # x may be the skip window of a concatenation buffer
        y.needs_grad = x.needs_grad
        if self.taping and y.needs_grad:
            def bwd():
                # gradient of the skip activation = pooled path + its half of the concatenation buffer's gradient
                skip = cat.grad if cat is not None else None
                if y.grad is None:
                    return
                dx = self._new(n, h, w, x.c)
                ops.maxpool2_bwd(x.t, y.grad, dx, dskip=skip, c=x.c, x_coff=x.coff, dskip_coff=x.coff)
                x.set_grad(dx)
            self.tape.append(bwd)
        return y

    def _up_bwd(self, t, up, cat):
        """backward of the ConvTranspose2d scatter: un-shuffle the 'up' half of the concatenation gradient into the
        4*Cout-row layout of the 1x1 layer; its per-channel sum is the ConvTranspose2d bias gradient."""
        g = cat.grad
        if g is None:
            return
        n, h2, w2, ld = g.shape
        c = up.c_up
        if up.mod.bias.requires_grad:
            self._grad_slot("bT", up).copy_(self._colsum(g, n * h2 * w2, c, coff=c)[0])
        dt = torch.empty_like(t.t)
        ops.pixel_unshuffle2(g, dt, c, src_coff=c)
        t.set_grad(dt)

    def _unet_fwd(self, U, x):
        """blocks/unet.py:49-95.  x [B,H,W,16] (3 real channels) -> masks [B,H,W,r16(num_classes)] (+ bottleneck)."""
        self.training = U["mod"].training
        n, H, W, _ = x.t.shape
        if H % 16 or W % 16:
            raise NotImplementedError("UNet on the HIP path needs H and W divisible by 16 (no output_padding rows in "
                                      "the transposed convolutions); the reference configs use 224/256")
        cats, h = [], x
        hh, ww = H, W
        for i in range(4):
            c = U["dwn"][i]["c2"].cout
            cat = Var(self._new(n, hh, ww, 2 * c))          # torch.cat([x_k, up], 1) buffer (unet.py:72)
            # the block's last BatchNorm writes the skip half directly -- and, fused, the pooled tensor of the next level
            # (train mode or taped: the eval-mode fold has no BatchNorm pass to fuse into)
            fused = (self._new(n, hh // 2, ww // 2, c) if self.fuse_bn_pool and (self.training or self.taping or not self.fold_bn_eval)
                     else None)
            a = self._conv3(h, U["dwn"][i], out=cat, pool_to=fused)
            cats.append(cat)
            h = self._maxpool2(a, cat, fused=fused)
            hh, ww = hh // 2, ww // 2
        # round 4 (late): the last BatchNorm + ReLU of a block whose one consumer is a 1x1 layer (the transposed convolutions, the
        # final classifier) stays pending on its pre-activation and is applied on load by that launch (untaped forward: _last_bn)
        lazy = self.fuse_in_bn_1x1 and not self.taping and not getattr(self, "return_inter", False)
        x5 = h = self._conv3(h, U["dwn"][4], defer=lazy)
        for j in range(4):
            cat, up = cats[3 - j], U["up"][j]
            if not self.taping and self.fuse_upconv_shuffle and self._upconv_fused(h, up, cat):
                h = self._conv3(cat, U["up_forw"][j], defer=lazy)
                continue
            h = self._materialize(h)
            t = self._conv(h, up, bias=True)                                  # [n, h, w, 4*Cout]
            ops.pixel_shuffle2(t.t, cat.t, up.c_up, dst_coff=up.c_up)
            if self.taping and t.needs_grad:
                cat.needs_grad = True
                self.tape.append(lambda t=t, up=up, cat=cat: self._up_bwd(t, up, cat))
            h = self._conv3(cat, U["up_forw"][j], defer=lazy)
        return self._conv1x1_after_bn(h, U["out"]), (None if x5.pending_bn is not None else x5)

    fuse_in_bn_1x1 = os.environ.get("PMOE_PUNET_BN_1X1", "1") != "0"      # round 4 (late): see _unet_fwd (the variable: A/B runs)
    fuse_upconv_shuffle = True     # round 4: ConvTranspose2d = 1x1 GEMM whose store scatters the 2x2 blocks itself (frozen / untaped path)

    def _upconv_fused(self, h, up, cat):
        """ConvTranspose2d(k2, s2) + its half of torch.cat in ONE launch: the 1x1 direct kernel writes channel (2 dy + dx) c_up + c
        of pixel (y, x) to cat[2 y + dy, 2 x + dx, c_up + c] (pmoe_conv_desc.shuffle_c) -- no [n, h, w, 4 c_up] intermediate, no
        pixel-shuffle launch.  False: this shape is not served (small maps, sides that are not powers of two): the caller runs the pair."""
        kw = dict(cin=up.cinp, cout=up.cout_st, coutp=up.coutp, ipe=self.B, ks=1, stride=1, pad=0, in_coff=h.coff,
                  out_coff=up.c_up, bias=up.bias_packed, shuffle2_c=up.c_up)
        if h.t.dtype != torch.bfloat16 or up.cout_st != 4 * up.c_up:
            return False
        if h.pending_bn is not None:                      # (its BatchNorm + ReLU applied on load: conv1x1_direct_kernel<MT, true>)
            kw.update(res_mode=hip.RES_INBN, bn_coef=h.pending_bn[0])
            if ops.conv2d(h.t, up.w_fwd, cat.t, plan_only=True, **kw) not in (1462, 1464):
                return False
        elif ops.conv2d(h.t, up.w_fwd, cat.t, plan_only=True, **kw) not in (1452, 1454):
            return False
        ops.set_meta(flop=2.0 * h.t.shape[0] * h.t.shape[1] * h.t.shape[2] * up.cout * up.cin, name=up.name + "+shuffle")
        ops.conv2d(h.t, up.w_fwd, cat.t, **kw)
        return True

    fold_entry_eca = True  # round 4: the entry block's two ECA gates folded into per-image weights of the convolutions they feed

    def _entry_fwd(self, masks):
        eb = self.entry
        self.training = self.pu.entry_block.training
        n, h, w, _ = masks.t.shape
        if self.fold_entry_eca and self.fold_eca_gate and self.training and not self.taping and h * w >= 256:
            # untaped train-mode forward (the frozen PU-Net inside a training step): conv(x * g[n]) = conv(x, W * g[n]) -- the two
            # gated activations (92 and 64 channels at full resolution) are never written, and the second gate's average pool comes
            # from the BatchNorm pass that writes its input (basics.py:79-134 via model/punet.py:60-68)
            z1, st1 = self._eca_conv_folded(masks, eb["eca1"], eb["conv1"])
            a = self._bn(z1, eb["bn1"], relu=True, stats=st1, want_gap=True)
            z, _ = self._eca_conv_folded(a, eb["eca2"], eb["conv2"])
            return self._bn(z, eb["bn2"], relu=True)        # 3 real channels: centred colstats pass (no fused epilogue stats)
        a = self._eca(masks, eb["eca1"], shared=False)
        a = self._conv_bn(a, eb["conv1"], eb["bn1"], relu=True)
        a = self._eca(a, eb["eca2"], shared=False)
        z = self._conv(a, eb["conv2"], bias=False)       # 3 real channels: centred colstats pass (no fused epilogue stats)
        return self._bn(z, eb["bn2"], relu=True)

    def _cat_masks(self, srcs, dst, nc):
        """torch.cat of class masks along channels (zero padded to the 16-wide row): one gather launch for up to 8 masks."""
        if len(srcs) <= 8:
            ops.cat_windows(srcs, dst, nc)
            return
        dst.zero_()
        for k, t in enumerate(srcs):
            ops.copy_window(t, 0, dst, k * nc, nc)

    def _punet_fwd(self, images):
        """punet.py:75-120: T past frames through ``unet``, then F autoregressive steps of
        cat(4 masks) -> entry_block -> pred_unet.  Returns x0 [B,H,W,r16(F*classes)] or the bottleneck feature."""
        pu = self.pu
        Bsz, T = images.shape[0], images.shape[1]
        if T != pu.n_past_frames:
            raise AssertionError("Number of images should match number of past frames")      # punet.py:84-86
        if pu.n_future_frames == 0 and not self._punet_trains:
            raise NotImplementedError("PUNetExpert needs future_frames > 0 (moe.py:286-289 sizes its stem from it)")
        H, W = images.shape[-2:]
        nc, cpad = pu.num_classes, r16(pu.in_features)
        masks = []
        # tests/punet_parity.py (per-pass teacher forcing): ``debug_pass_out`` (a list) collects the mask tensor every U-Net
        # pass wrote; ``debug_forced_masks`` (T + F tensors [B,classes,H,W]) REPLACES each pass's output by the given mask
        # before the later passes read it, so that every pass runs on the checker's inputs and errors do not compound
        kept, forced = getattr(self, "debug_pass_out", None), getattr(self, "debug_forced_masks", None)

        def _pass_done(out):
            if kept is not None:
                kept.append(out.t)
            if forced is None:
                return out
            given = Var(torch.empty_like(out.t))
            ops.nchw_to_nhwc(forced[len(masks)].to(self.dev).contiguous().float(), given.t)
            return given
        for i in range(T):
            xi = Var(self._new(Bsz, H, W, cpad))
            ops.nchw_to_nhwc(images[:, i].contiguous().float(), xi.t)
            out = self._unet_fwd(self.unet, xi)[0]
            masks.append(_pass_done(out))
        F_ = pu.n_future_frames
        if F_ == 0:                                    # punet.py:91-96: segmentation of the current frame
            self._pred_masks = None
            return masks[-1], None
        inter = None
        for f in range(F_):
            if self.taping:
                self._step_begin(f)
            cat = Var(self._new(Bsz, H, W, r16(T * nc)))
            srcs = masks[-T:]
            self._cat_masks([m.t for m in srcs], cat.t, nc)
            cat.needs_grad = any(m.needs_grad for m in srcs)
            if self.taping and cat.needs_grad:
                # gradient of torch.cat (punet.py:104,113): each window is ADDED to its mask's gradient (a predicted mask
                # feeds up to T later steps and the loss)
                def cat_bwd(cat=cat, srcs=srcs):
                    if cat.grad is None:
                        return
                    for k, m in enumerate(srcs):
                        if m.needs_grad:
                            ops.add_window(cat.grad, k * nc, m.grad, 0, nc)
                self.tape.append(cat_bwd)
            e = self._entry_fwd(cat)
            m, inter = self._unet_fwd(self.pred_unet, e)
            if not self.taping:
                m = _pass_done(m)
            masks.append(m)
        x0 = None
        if not self.return_inter:                  # torch.stack(outs,1).view(B,-1,H,W)  (punet.py:120, moe.py:311)
            x0 = Var(self._new(Bsz, H, W, self._x0_width(F_ * nc)))
            self._cat_masks([m.t for m in masks[T:]], x0.t, nc)
        self._pred_masks = masks[T:] if self.taping else None
        return x0, inter

    def _step_begin(self, f):
        """hook: start of autoregressive step ``f`` while taping (the stage-1 engine accumulates shared-weight gradients)."""

    # ------------------------------------------------------------------ network
    def forward(self, images, speed, command, training, taping, dtype, base_seed=0):
        """-> actions [B,2] (tanh), pred_speed [B,1], state."""
        for p in self.pu.parameters():
            if p.requires_grad:
                raise NotImplementedError("PU-Net parameters must stay frozen on the HIP path (PUNetExpert freezes them, "
                                          "moe.py:280); stage-1 PU-Net training is SURVEY.md section 8f N4")
        Bsz = self._begin(images, training, taping, dtype, base_seed)
        self._shadows_in()
        e = self.experts[0]
        top_training = training
        spd, cmd = self._measurement_inputs(speed, command)
        taping_saved, self.taping = self.taping, False          # nothing inside the frozen PU-Net is taped
        if getattr(self, "debug_x0", None) is not None and not self.return_inter:
            # tests/punet_parity.py (teacher forcing): the predicted masks [B,F,classes,H,W] are GIVEN, the frozen PU-Net is
            # skipped -- the trainable half (138-channel stem, ResNet, heads) is then compared on identical inputs, without
            # the chained train-mode U-Nets' sensitivity in the loop
            mk = self.debug_x0
            Hm, Wm = mk.shape[-2:]
            x0 = Var(self._new(Bsz, Hm, Wm, self._x0_width(mk.shape[1] * mk.shape[2])))
            ops.nchw_to_nhwc(mk.reshape(Bsz, -1, Hm, Wm).contiguous().float(), x0.t)
            inter = None
        else:
            x0, inter = self._punet_fwd(images)
        if getattr(self, "debug_keep_x0", False):
            self.debug_x0_kept = x0.t if x0 is not None else None
        self.taping = taping_saved
        self.training = top_training
        feat = Var(self._new(self.N, 1, 1, 1536))
        if self.return_inter:
            self._gap_to(inter, feat, 0)
        else:
            self.training = e.backbone.training
            self._backbone_fwd(x0, feat)
        self.training = e.speed_encoder.training
        self._mlp(spd, self.speed_enc, out=feat, out_coff=512, in_shared=True)
        self._mlp(cmd, self.cmd_enc, out=feat, out_coff=1024, in_shared=True)
        sp = self._mlp(feat, self.speed_pred)
        af = self._mlp(feat, self.action_feat)
        head = self._conv(af, self.head)
        actions = torch.empty(Bsz, 2, dtype=F32, device=self.dev)
        speeds = torch.empty(Bsz, 1, dtype=F32, device=self.dev)
        ops.action_head_fwd(head.t.view(Bsz, -1), sp.t.view(Bsz, -1), actions, speeds, Bsz)
        self._shadows_out()
        self._bump_batch_counters()
        state = dict(tape=self.tape, tail=(head, sp, actions), B=self.B, N=self.N, dev=self.dev, dtype=self.dtype)
        self.tape = None
        return actions, speeds, state

    def _tail_bwd(self, tail, dactions, dspeeds):
        head, sp, actions = tail
        dhead = torch.empty_like(head.t)
        dspd = torch.empty_like(sp.t)

        def c(t):
            return t.contiguous().float() if t is not None else None
        ops.action_head_bwd(actions, c(dactions), c(dspeeds), dhead.view(self.B, -1), dspd.view(self.B, -1), self.B)
        if dactions is not None:
            head.set_grad(dhead)
        if dspeeds is not None:
            sp.set_grad(dspd)


class PredictiveUnetEngine(PUNetEngine):
    """``PredictiveUnet`` on its own (``model/punet.py:75-120``): stage-1 training (``trainer/train_1.py:129-141``,
    SURVEY.md section 8f N4) and plain segmentation-forecast inference.

    ``entry_block`` and ``pred_unet`` train, ``unet`` stays frozen (punet.py:46-48).  The autoregressive loop applies
    the SAME weights ``future_frames`` times, so backward is back-propagation through time: every step's closures write
    the gradient arena, and a closure recorded at the start of each step adds the arena into an accumulator and clears it
    (the last one to run adds the accumulator back), so the layer backward code stays "write, don't accumulate".
    A predicted mask receives gradient from the loss and from up to ``past_frames`` later steps through the channel
    concatenation (``add_window``)."""
    _punet_trains = True

    def __init__(self, punet):
        import types
        self.return_inter = bool(punet.inter_repr)
        ExpertGroupEngine.__init__(self, [types.SimpleNamespace(punet=punet)], alt=False)

    def _collect_network(self, ex):
        self._collect_pre_backbone(ex)
        self.conv1 = self.eca1 = self.head = None
        self.blocks = []
        for blk in self.unet["dwn"] + self.unet["up_forw"]:          # frozen, fed by images: no data gradients
            blk["c1"].need_dgrad = blk["c2"].need_dgrad = False
        self.unet["out"].need_dgrad = False
        for up in self.unet["up"]:
            up.need_dgrad = False

    def _step_begin(self, f):
        first = f == 0

        def flush():
            n = self._arena_numel
            if first:
                ops.add_window(self._acc.view(1, n), 0, self._arena.view(1, n), 0, n)
                self._acc = None
                self._accum_done = True
            else:
                ops.add_window(self._arena.view(1, n), 0, self._acc.view(1, n), 0, n)
                self._arena.zero_()
        self.tape.append(flush)

    def _final_prefix(self):
        # slots are final only after the last accumulation (data-parallel buckets all fly at the end of backward)
        return super()._final_prefix() if self._accum_done else 0

    def forward(self, images, training, taping, dtype):
        """images [B,T,C,H,W] f32 -> logits [B,F,classes,H,W] f32 (or the bottleneck feature [B,512] with ``inter_repr``)."""
        pu = self.pu
        if pu.unet_inter_repr:
            raise NotImplementedError("PredictiveUnet(unet_inter_repr=True) is never configured by the reference "
                                      "(punet.py:33-39 passes the default)")
        if taping and self.return_inter:
            raise NotImplementedError("PredictiveUnet(inter_repr=True) is inference-only (punet.py:99: 'not suitable for "
                                      "training')")
        for p in pu.unet.parameters():
            if p.requires_grad:
                raise NotImplementedError("PredictiveUnet.unet must stay frozen (punet.py:46-48); stage-0 U-Net training is "
                                          "out of scope (SURVEY.md section 8)")
        Bsz = self._begin(images, training, taping, dtype, 0)
        self._shadows_in()
        x0, inter = self._punet_fwd(images)
        self._shadows_out()
        self._bump_batch_counters()
        H, W = images.shape[-2:]
        if self.return_inter:
            feat = Var(self._new(Bsz, 1, 1, inter.c))
            self._gap_to(inter, feat, 0)
            out = feat.t.view(Bsz, -1).float()
        else:
            F_, nc = pu.n_future_frames, pu.num_classes
            out = torch.empty(Bsz, max(F_, 1) * nc, H, W, dtype=F32, device=self.dev)
            ops.nhwc_to_nchw(x0.t, out, max(F_, 1) * nc)
            if F_ > 0:
                out = out.view(Bsz, F_, nc, H, W)
        state = dict(tape=self.tape, tail=self._pred_masks, B=self.B, N=self.N, dev=self.dev, dtype=self.dtype)
        self.tape = self._pred_masks = None
        return out, state

    def _tail_bwd(self, masks, dout):
        """d loss / d logits [B,F,classes,H,W] f32 -> the initial gradient of every predicted mask (NHWC, zero padded)."""
        if dout is None or not masks:
            return
        Bsz, F_, nc, H, W = dout.shape
        g = self._new(Bsz, H, W, r16(F_ * nc))
        ops.nchw_to_nhwc(dout.contiguous().float().view(Bsz, F_ * nc, H, W), g)
        for f, m in enumerate(masks):
            mg = torch.zeros_like(m.t)
            ops.copy_window(g, f * nc, mg, 0, nc)
            m.set_grad(mg)

    def backward(self, tape_state, *douts):
        self._accum_done = False
        self._layout_arena()                             # (the accumulator has the arena's padded length)
        self._acc = torch.zeros(self._arena_numel, dtype=F32, device=tape_state["dev"])
        return super().backward(tape_state, *douts)
