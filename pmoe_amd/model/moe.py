"""Drop-in for ``PMoE/model/moe.py``: ``get_model(cfg)``, ``MixtureOfExperts``, ``BaseExpert`` ...

Same constructor arguments (any attribute-style mapping, e.g. an OmegaConf node), same
``forward(images, speed, command)`` / ``sample(...)`` signatures and return types, same
``state_dict`` keys, so ``trainer/train_2.py`` and ``autoagents/image_agent.py``
can import it instead of the reference module.  All arithmetic runs in the grouped HIP engine.
"""
import torch
import torch.nn as nn
import torch.distributions as D

from .. import ops
from ..engine import ExpertGroupEngine
from ..utils import freeze  # noqa: F401  (re-exported: the reference imports it from utils.nn)
from . import blocks as B
from .punet import PredictiveUnet

_DEFAULT_DTYPE = torch.bfloat16


def set_default_compute_dtype(dtype):
    """bf16 (default; BASELINE config) or float32 (exact-f32 MFMA path, used for 1e-4 parity)."""
    global _DEFAULT_DTYPE
    if dtype not in (torch.bfloat16, torch.float32):
        raise ValueError("compute dtype must be torch.bfloat16 or torch.float32")
    _DEFAULT_DTYPE = dtype


def get_model(cfg):
    """``moe.py:25-47``."""
    model_type = cfg.type
    assert model_type is not None, "Network type can not be None"
    if model_type in ["moe", "moe_alt"]:
        return MixtureOfExperts(cfg)
    elif model_type == "moe_shared":
        return MixtureOfExpertsShared(cfg)
    elif model_type in ["punet", "punet_inter"]:
        return PUNetExpert(cfg)
    elif model_type in ["pmoe", "pmoe+pretrained"]:
        assert cfg.pmoe.moe_dir != "", "MoE pretrained weights directory should be specified"
        if model_type == "pmoe+pretrained":
            assert cfg.pmoe.punet_dir != "", "PU-Net pretrained weights directory should be specified"
        return PMoE(cfg)
    else:
        raise ValueError(
            f"{model_type} is UNKNOWN, model type should be one of 'moe', 'punet', "
            f"'punet_inter', 'pmoe', 'pmoe+pretrained', 'moe_alt'")


class _GroupFn(torch.autograd.Function):
    """The whole grouped network as one autograd node (inputs: the flat parameter list)."""

    @staticmethod
    def forward(ctx, engine, images, speed, command, training, dtype, seed, taping, *params):
        *outs, state = engine.forward(images, speed, command, training, taping, dtype, seed)
        ctx.engine, ctx.state = engine, state
        ctx.set_materialize_grads(False)      # an unused output (e.g. pred_speed under pmoe_loss) leaves its head's .grad None
        ctx.param_ids = [id(p) for p in params]
        return tuple(outs)       # (probs, mean, std, speeds) for the mixtures; (actions, pred_speed) for PUNetExpert

    @staticmethod
    def backward(ctx, *douts):
        grads = ctx.engine.backward(ctx.state, *douts)
        ctx.state = None
        out = [grads.get(i) if need else None for i, need in zip(ctx.param_ids, ctx.needs_input_grad[8:])]
        return (None,) * 8 + tuple(out)


class _Grouped(nn.Module):
    """Shared forward plumbing of modules that own a list of experts."""

    compute_dtype = None      # None -> module-level default (bf16)
    fp8_weights = False       # True (with bf16 compute): BASELINE config 5 -- the ResNet layer1-4 forward convolutions run on
                              # e4m3 weights / e4m3 activations and the fp8 matrix cores (pmoe_conv_desc.w_fp8)

    def _engine(self):
        eng = self.__dict__.get("_eng")
        if eng is None:
            eng = self._make_engine()
            self.__dict__["_eng"] = eng          # not a submodule / not in state_dict / rebuilt after deepcopy
        return eng

    def _shared_k(self):
        return 0

    def _make_engine(self):
        return ExpertGroupEngine(self._expert_list(), alt=self._alt(), shared_k=self._shared_k())

    def __deepcopy__(self, memo):
        # AveragedModel(model) deep-copies (train_2.py:120): drop the engine (raw device buffers), copy the rest
        eng = self.__dict__.pop("_eng", None)
        try:
            cls = self.__class__
            new = cls.__new__(cls)
            memo[id(self)] = new
            import copy
            for k, v in self.__dict__.items():
                new.__dict__[k] = copy.deepcopy(v, memo)
        finally:
            if eng is not None:
                self.__dict__["_eng"] = eng
        return new

    def _run(self, images, speed, command):
        eng = self._engine()
        dtype = self.compute_dtype or _DEFAULT_DTYPE
        eng.fp8 = bool(self.fp8_weights) and dtype == torch.bfloat16
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if self.training else 0
        # grad mode is off inside Function.forward, so decide here whether a backward tape is needed
        taping = torch.is_grad_enabled() and any(p.requires_grad for p in eng.flat_params)
        return _GroupFn.apply(eng, images, speed, command, self.training, dtype, seed, taping, *eng.flat_params)

    def enable_data_parallel(self, group=None, n_buckets=6, always=False):
        """Average parameter gradients over ``group`` (default WORLD) inside backward, bucketed and
        overlapped (pmoe_amd.parallel.BucketedAllReduce).  ``always``: issue the collectives even in a one-rank group
        (exercises the RCCL path on a single GPU)."""
        eng = self._engine()
        eng.dp_group, eng.dp_enabled, eng.dp_buckets, eng.dp_always = group, True, n_buckets, always
        return self


class BaseExpert(_Grouped):
    """``moe.py:50-101``: one expert.  Runs as a group of one when called on its own."""

    def __init__(self, params):
        super().__init__()
        self.speed_encoder = B.make_mlp(**params.speed_encoder)
        self.command_encoder = B.make_mlp(**params.command_encoder)
        if params.backbone.type != "rgb":
            raise NotImplementedError("backbone.type 'segmentation' (get_unet) is not built: the reference cannot run it either -- "
                                      "Sequential(entry_block, UNet(inter_repr=True)) returns a TUPLE (blocks/unet.py:92-95) that "
                                      "BaseExpert.forward's torch.cat rejects (model/moe.py:92-95); SURVEY.md section 2 #3")
        self.backbone = B.get_backbone(**{**params.backbone.rgb, "n_frames": params.backbone.n_frames})
        self.speed_pred = B.make_mlp(**params.speed_prediction)
        self.action_features = B.make_mlp(**params.action_head)
        width = params.action_head.dims[-1]
        self.alpha = B.Linear(width, 1)
        self.action_pred = B.Linear(width, 4)

    def _expert_list(self):
        return [self]

    def _alt(self):
        return False

    def _make_engine(self):
        eng = super()._make_engine()
        eng.raw_alpha = True        # a lone expert has no group to normalise over: the gate kernel hands back alpha itself
        return eng

    def forward(self, images, speed, command):
        """-> alpha [B,1], mean [B,2], std [B,2], pred_speed [B,1] (moe.py:74-101; BaseExpertAlt: moe.py:112-128).
        Runs as a group of ONE through the same engine the mixture uses (MixtureOfExperts builds its own grouped engine
        over all its experts and never calls this)."""
        alpha, mean, std, speeds = self._run(images, speed, command)
        return alpha, mean[:, 0], std[:, 0], speeds[:, 0]


class BaseExpertAlt(BaseExpert):
    """``moe.py:104-128``: alpha = MLP(1536 -> 512 -> 1) on the concatenated features, no ReLU."""

    def __init__(self, params):
        super().__init__(params)
        self.alpha = B._Seq(B.Linear(1536, 512), B.Activation("relu"), B.Linear(512, 1))

    def _alt(self):
        return True


class MixtureOfExperts(_Grouped):
    """``moe.py:131-177``."""

    def __init__(self, params):
        super().__init__()
        self.k = params.n_experts
        self._is_alt = params.type != "moe"
        base = BaseExpert if params.type == "moe" else BaseExpertAlt
        self.moe = nn.ModuleList([base(params) for _ in range(self.k)])

    def _expert_list(self):
        return list(self.moe)

    def _alt(self):
        return self._is_alt

    def mixture_params(self, images, speed, command):
        """probs [B,E], mean [B,E,2], std [B,E,2], speeds [B,E,1]: the deterministic tensors behind the
        distribution (gate softmax of moe.py:150-151 included)."""
        return self._run(images, speed, command)

    def forward(self, images, speed, command):
        probs, mean, std, speeds = self._run(images, speed, command)
        dist = MixtureDistribution(probs, mean, std)
        return dist, speeds

    def sample(self, images, speed, command):
        probs, mean, std, _ = self._run(images, speed, command)
        return MixtureDistribution(probs, mean, std).sample()


class MixtureOfExpertsShared(_Grouped):
    """``moe.py:180-265``: ONE trunk (encoders, backbone, speed / action-feature heads) whose last layer emits
    ``n_experts`` Gaussian components: ``action_pred = Linear(512, 4*n_experts)`` viewed as [B, n_experts, 4] and
    ``alpha = Linear(512, n_experts)`` (plain softmax).  ``forward`` returns (distribution, pred_speed [B,1])."""

    def __init__(self, params):
        super().__init__()
        self.speed_encoder = B.make_mlp(**params.speed_encoder)
        self.command_encoder = B.make_mlp(**params.command_encoder)
        if params.backbone.type != "rgb":
            raise NotImplementedError("backbone.type 'segmentation' (get_unet) is not built: the reference cannot run it either -- "
                                      "Sequential(entry_block, UNet(inter_repr=True)) returns a TUPLE (blocks/unet.py:92-95) that "
                                      "BaseExpert.forward's torch.cat rejects (model/moe.py:92-95); SURVEY.md section 2 #3")
        self.backbone = B.get_backbone(**{**params.backbone.rgb, "n_frames": params.backbone.n_frames})
        self.speed_pred = B.make_mlp(**params.speed_prediction)
        self.action_features = B.make_mlp(**params.action_head)
        width = params.action_head.dims[-1]
        self.n_experts = params.n_experts
        self.alpha = B.Linear(width, params.n_experts)
        self.action_pred = B.Linear(width, 4 * params.n_experts)

    def _expert_list(self):
        return [self]

    def _alt(self):
        return False

    def _shared_k(self):
        return self.n_experts

    def mixture_params(self, images, speed, command):
        """probs [B,K], mean [B,K,2], std [B,K,2], pred_speed [B,1]."""
        return self._run(images, speed, command)

    def forward(self, images, speed, command):
        probs, mean, std, speeds = self._run(images, speed, command)
        return MixtureDistribution(probs, mean, std), speeds

    def sample(self, images, speed, command):
        probs, mean, std, _ = self._run(images, speed, command)
        return MixtureDistribution(probs, mean, std).sample()


class PUNetExpert(_Grouped):
    """``moe.py:268-323``: frozen PU-Net (future segmentation masks) -> ResNet18-ECA backbone over the
    ``future_frames * num_classes`` mask channels -> ``tanh`` action head; ``punet_inter`` feeds the PU-Net bottleneck
    vector instead of a backbone.  Loads ``params.punet_path`` (key ``"model"``) exactly like the reference."""

    def __init__(self, params):
        super().__init__()
        self.return_inter = True if params.type == "punet_inter" else False
        params.punet.inter_repr = self.return_inter
        self.speed_encoder = B.make_mlp(**params.speed_encoder)
        self.command_encoder = B.make_mlp(**params.command_encoder)
        self.punet = PredictiveUnet(**params.punet)
        punet_weights = torch.load(params.punet_path, map_location=params.device)
        self.punet.load_state_dict(punet_weights["model"])
        self.punet = freeze(self.punet)
        self.backbone = None if self.return_inter else B.get_backbone(
            **{**params.backbone.rgb, "n_frames": params.punet.future_frames, "n_channels": params.punet.num_classes})
        self.speed_pred = B.make_mlp(**params.speed_prediction)
        self.action_pred = B._Seq(B.make_mlp(**params.action_head), B.Linear(params.action_head.dims[-1], 2))

    def _make_engine(self):
        from ..engine_punet import PUNetEngine
        return PUNetEngine(self)

    def forward(self, images, speed, command):
        """-> (actions [B,2] in (-1,1), pred_speed [B,1])."""
        return self._run(images, speed, command)

    def sample(self, images, speed, command):
        return self.forward(images, speed, command)[0]


class _BlendFn(torch.autograd.Function):
    """``tanh(cat(lat_weights([moe_x, punet_x]), long_weights([moe_y, punet_y])))`` (moe.py:353-356)."""

    @staticmethod
    def forward(ctx, moe_act, pu_act, lat_w, lat_b, long_w, long_b):
        Bsz = moe_act.shape[0]
        moe_act, pu_act = moe_act.contiguous().float(), pu_act.contiguous().float()
        out = torch.empty(Bsz, 2, dtype=torch.float32, device=moe_act.device)
        ops.blend_fwd(moe_act, pu_act, lat_w.contiguous(), lat_b.contiguous(), long_w.contiguous(), long_b.contiguous(),
                      out, Bsz)
        ctx.save_for_backward(moe_act, pu_act, lat_w, long_w, out)
        return out

    @staticmethod
    def backward(ctx, dout):
        moe_act, pu_act, lat_w, long_w, out = ctx.saved_tensors
        Bsz = moe_act.shape[0]
        dlw, dgw = torch.empty_like(lat_w), torch.empty_like(long_w)
        dlb = torch.empty(1, dtype=torch.float32, device=out.device)
        dgb = torch.empty(1, dtype=torch.float32, device=out.device)
        dpu = torch.empty_like(pu_act) if ctx.needs_input_grad[1] else None
        ops.blend_bwd(moe_act, pu_act, lat_w.contiguous(), long_w.contiguous(), out, dout.contiguous().float(),
                      dlw, dlb, dgw, dgb, dpu, Bsz)
        return None, dpu, dlw, dlb, dgw, dgb


class PMoE(nn.Module):
    """``moe.py:326-363``: frozen mixture of experts + PU-Net expert, blended per axis by two ``Linear(2,1)``.
    ``forward`` returns ``(actions [B,2], -1)`` and is stochastic (``dists.sample()``), like the reference."""

    def __init__(self, params):
        super().__init__()
        assert params.pmoe.moe_dir is not None, "MoE weights should be provided"
        self.moe = MixtureOfExperts(params)
        # SWA checkpoints carry extra keys, therefore strict=False (moe.py:336-337)
        self.moe.load_state_dict(torch.load(params.pmoe.moe_dir, map_location="cpu"), strict=False)
        self.moe = freeze(self.moe, params.exclude_freeze, params.verbose)
        self.punet = PUNetExpert(params)
        if params.pmoe.punet_dir:
            self.punet.load_state_dict(torch.load(params.pmoe.punet_dir, map_location="cpu"), strict=False)
            self.punet = freeze(self.punet, params.exclude_freeze, params.verbose)
        self.lat_weights = B.Linear(2, 1)
        self.long_weights = B.Linear(2, 1)

    @property
    def compute_dtype(self):
        return self.moe.compute_dtype

    @compute_dtype.setter
    def compute_dtype(self, dtype):
        self.moe.compute_dtype = dtype
        self.punet.compute_dtype = dtype

    def blend(self, moe_actions, punet_actions):
        """the deterministic tail of ``forward`` (moe.py:353-356) on given per-model actions."""
        return _BlendFn.apply(moe_actions, punet_actions, self.lat_weights.weight, self.lat_weights.bias,
                              self.long_weights.weight, self.long_weights.bias)

    def forward(self, images, speed, command):
        punet_actions, _ = self.punet(images, speed, command)
        dists, _ = self.moe(images, speed, command)
        moe_actions = dists.sample()
        # -1 is the reference's dummy speed prediction (interface consistency)
        return self.blend(moe_actions, punet_actions), -1

    def sample(self, images, speed, command):
        return self.forward(images, speed, command)[0]


class MixtureDistribution(D.MixtureSameFamily):
    """The very distribution ``moe.py:154-156`` builds; also keeps the raw parameter tensors so that
    ``pmoe_amd.loss.moe_loss`` can use the fused HIP loss kernel."""

    def __init__(self, probs, mean, std):
        self.hip_params = (probs, mean, std)
        super().__init__(D.Categorical(probs), D.Independent(D.Normal(mean, std), 1), validate_args=False)
