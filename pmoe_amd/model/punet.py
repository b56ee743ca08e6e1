"""Drop-in parameter container for ``PMoE/model/punet.py`` (``PredictiveUnet``).

Same constructor arguments and ``state_dict`` keys (``unet.*``, ``entry_block.*``, ``pred_unet.*``); like the
reference it loads the stage-0 U-Net checkpoint ``torch.load(model_path)[model_name]`` with ``strict=False`` and
freezes it (punet.py:40-55).  Inside ``PUNetExpert`` the arithmetic is issued by ``pmoe_amd.engine_punet.PUNetEngine``
(stage 2: the whole PU-Net frozen).  Called on its own -- ``PredictiveUnet.forward(img_list)``, the stage-1 trainer's
``self.model(img)`` (trainer/train_1.py:131-134) -- it runs on ``PredictiveUnetEngine``: forward, and backward through the
autoregressive loop for ``entry_block`` / ``pred_unet``, as one autograd node.
"""
import copy

import torch

from . import blocks as B


class _PUNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, engine, images, training, dtype, taping, *params):
        out, state = engine.forward(images, training, taping, dtype)
        ctx.engine, ctx.state = engine, state
        ctx.param_ids = [id(p) for p in params]
        return out

    @staticmethod
    def backward(ctx, dout):
        grads = ctx.engine.backward(ctx.state, dout)
        ctx.state = None
        out = [grads.get(i) if need else None for i, need in zip(ctx.param_ids, ctx.needs_input_grad[5:])]
        return (None,) * 5 + tuple(out)


class PredictiveUnet(B._Held):
    def __init__(self, past_frames=4, future_frames=4, in_features=3, num_classes=23, gamma=2, b=1, inter_repr=False,
                 unet_inter_repr=False, model_name="unet-swa", model_path="unet.pth"):
        super().__init__()
        self.n_past_frames = past_frames
        self.n_future_frames = future_frames
        self.inter_repr = inter_repr
        self.unet_inter_repr = unet_inter_repr
        self.in_features, self.num_classes = in_features, num_classes
        self.unet = B.UNet(in_features=in_features, out_features=num_classes, gamma=gamma, b=b, inter_repr=unet_inter_repr)
        checkpoint = torch.load(model_path, map_location="cpu")       # punet.py:40 (raises like the reference if absent)
        self.unet.load_state_dict(checkpoint[model_name], strict=False)
        for p in self.unet.parameters():
            p.requires_grad = False
        self.unet.eval()
        self.entry_block = B.EfficientConvBlock(in_ch=past_frames * num_classes, out_ch=in_features, gamma=gamma, b=b)
        self.pred_unet = B.UNet(in_features=in_features, out_features=num_classes, gamma=gamma, b=b, inter_repr=inter_repr)

    compute_dtype = None      # None -> pmoe_amd.model.moe's module-level default (bf16)

    def _engine(self):
        eng = self.__dict__.get("_eng")
        if eng is None:
            from ..engine_punet import PredictiveUnetEngine
            eng = self.__dict__["_eng"] = PredictiveUnetEngine(self)     # not a submodule / not in state_dict
        return eng

    def __deepcopy__(self, memo):
        # AveragedModel(model) deep-copies (train_1.py:113): drop the engine (raw device buffers), copy the rest
        eng = self.__dict__.pop("_eng", None)
        try:
            new = self.__class__.__new__(self.__class__)
            memo[id(self)] = new
            for k, v in self.__dict__.items():
                new.__dict__[k] = copy.deepcopy(v, memo)
        finally:
            if eng is not None:
                self.__dict__["_eng"] = eng
        return new

    def enable_data_parallel(self, group=None, n_buckets=6):
        """Average parameter gradients over ``group`` (default WORLD) inside backward (pmoe_amd.parallel).  The roll-out
        accumulates shared-weight gradients until its first step has run, so every bucket flies at the end of backward."""
        eng = self._engine()
        eng.dp_group, eng.dp_enabled, eng.dp_buckets = group, True, n_buckets
        return self

    def forward(self, img_list):
        """``punet.py:75-120``: img_list [B,T,C,H,W] -> logits of the ``future_frames`` predicted masks [B,F,classes,H,W]
        (f32), or the bottleneck feature [B,512] when ``inter_repr`` (inference only)."""
        assert img_list.shape[-4] == self.n_past_frames, "Number of images should match number of past frames"
        from . import moe as _moe
        eng = self._engine()
        dtype = self.compute_dtype or _moe._DEFAULT_DTYPE
        taping = torch.is_grad_enabled() and any(p.requires_grad for p in eng.flat_params)
        return _PUNetFn.apply(eng, img_list, self.training, dtype, taping, *eng.flat_params)
