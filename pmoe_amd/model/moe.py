"""Drop-in for ``PMoE/model/moe.py``: ``get_model(cfg)``, ``MixtureOfExperts``, ``BaseExpert`` ...

Same constructor arguments (any attribute-style mapping, e.g. an OmegaConf node), same
``forward(images, speed, command)`` / ``sample(...)`` signatures and return types, same
``state_dict`` keys, so ``trainer/train_2.py`` and ``autoagents/image_agent.py``
can import it instead of the reference module.  All arithmetic runs in the grouped HIP engine.
"""
import torch
import torch.nn as nn
import torch.distributions as D

from ..engine import ExpertGroupEngine
from ..utils import freeze  # noqa: F401  (re-exported: the reference imports it from utils.nn)
from . import blocks as B

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
    elif model_type in ["punet", "punet_inter", "pmoe", "pmoe+pretrained"]:
        raise NotImplementedError(
            f"model type {model_type!r} is a reference option that is not on the MI355X path yet "
            "(SURVEY.md section 8: second tier / config 4); 'moe', 'moe_alt' and 'moe_shared' are")
    else:
        raise ValueError(
            f"{model_type} is UNKNOWN, model type should be one of 'moe', 'punet', "
            f"'punet_inter', 'pmoe', 'pmoe+pretrained', 'moe_alt'")


class _GroupFn(torch.autograd.Function):
    """The whole grouped network as one autograd node (inputs: the flat parameter list)."""

    @staticmethod
    def forward(ctx, engine, images, speed, command, training, dtype, seed, taping, *params):
        probs, mean, std, speeds, state = engine.forward(images, speed, command, training, taping, dtype, seed)
        ctx.engine, ctx.state = engine, state
        ctx.param_ids = [id(p) for p in params]
        return probs, mean, std, speeds

    @staticmethod
    def backward(ctx, dprobs, dmean, dstd, dspeeds):
        grads = ctx.engine.backward(ctx.state, dprobs, dmean, dstd, dspeeds)
        ctx.state = None
        out = [grads.get(i) if need else None for i, need in zip(ctx.param_ids, ctx.needs_input_grad[8:])]
        return (None,) * 8 + tuple(out)


class _Grouped(nn.Module):
    """Shared forward plumbing of modules that own a list of experts."""

    compute_dtype = None      # None -> module-level default (bf16)

    def _engine(self):
        eng = self.__dict__.get("_eng")
        if eng is None:
            eng = ExpertGroupEngine(self._expert_list(), alt=self._alt(), shared_k=self._shared_k())
            self.__dict__["_eng"] = eng          # not a submodule / not in state_dict / rebuilt after deepcopy
        return eng

    def _shared_k(self):
        return 0

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
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if self.training else 0
        # grad mode is off inside Function.forward, so decide here whether a backward tape is needed
        taping = torch.is_grad_enabled() and any(p.requires_grad for p in eng.flat_params)
        return _GroupFn.apply(eng, images, speed, command, self.training, dtype, seed, taping, *eng.flat_params)

    def enable_data_parallel(self, group=None, n_buckets=6):
        """Average parameter gradients over ``group`` (default WORLD) inside backward, bucketed and
        overlapped (pmoe_amd.parallel.BucketedAllReduce)."""
        eng = self._engine()
        eng.dp_group, eng.dp_enabled, eng.dp_buckets = group, True, n_buckets
        return self


class BaseExpert(_Grouped):
    """``moe.py:50-101``: one expert.  Runs as a group of one when called on its own."""

    def __init__(self, params):
        super().__init__()
        self.speed_encoder = B.make_mlp(**params.speed_encoder)
        self.command_encoder = B.make_mlp(**params.command_encoder)
        if params.backbone.type != "rgb":
            raise NotImplementedError("backbone.type 'segmentation' (get_unet) is not on the HIP path (SURVEY.md section 2 #3)")
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

    def forward(self, images, speed, command):
        """-> alpha [B,1], mean [B,2], std [B,2], pred_speed [B,1] (moe.py:74-101).  ``alpha`` is returned as
        log-probabilities' argument: for a single expert the softmax is trivially 1, so the raw
        (post-ReLU) coefficient is recovered from the head output."""
        raise RuntimeError("call the expert through MixtureOfExperts (grouped execution); a lone BaseExpert "
                           "has no defined mixture output on the HIP path")


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
            raise NotImplementedError("backbone.type 'segmentation' (get_unet) is not on the HIP path (SURVEY.md section 2 #3)")
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


class MixtureDistribution(D.MixtureSameFamily):
    """The very distribution ``moe.py:154-156`` builds; also keeps the raw parameter tensors so that
    ``pmoe_amd.loss.moe_loss`` can use the fused HIP loss kernel."""

    def __init__(self, probs, mean, std):
        self.hip_params = (probs, mean, std)
        super().__init__(D.Categorical(probs), D.Independent(D.Normal(mean, std), 1), validate_args=False)
