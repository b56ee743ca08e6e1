"""Closed-loop inference helper (reference caller ``autoagents/image_agent.py:127-177``).

At B=1 the eval-mode forward is ~150 small launches and the tick time is launch latency, not kernel time.  The engine
allocates only through torch's caching allocator and (in eval mode) never synchronises with the host, so the whole
chain can be captured ONCE into a HIP graph (``torch.cuda.CUDAGraph`` is hipGraph on ROCm) and replayed per tick with
new inputs copied into the captured input buffers.  Parameters are read through the captured packed-weight buffers; every
replay first compares the engine's build / pointer-table / version keys with those taken at capture and re-captures
(:meth:`GraphedMixture.refresh`) when a weight, a BatchNorm buffer, the compute dtype or the device has changed since.
"""
import torch

from . import hip
from .model.moe import MixtureDistribution


def _plan_key(model):
    """Everything a captured / recorded chain holds raw pointers into, beyond its private activation pool: the engine's
    packed weight banks and pointer tables (rebuilt when the compute dtype, the fp8 switch or the device changes, or when a
    parameter's storage moves), their contents (parameter versions) and the eval-mode BatchNorm folds (buffer versions).
    A replay whose key differs from the one taken at capture would read freed or stale memory."""
    eng = model._engine()
    bufs = [b for l in eng.all_bns for m in l.mods for b in (m.running_mean, m.running_var)]
    return (eng._built_for, eng._ptr_key, eng._packed_version, sum(p._version for p in eng.flat_params),
            sum(b._version for b in bufs), tuple(b.data_ptr() for b in bufs[:4]))


class GraphedMixture:
    """``gm = GraphedMixture(model, images, speed, command)`` captures ``model.mixture_params`` for inputs of that shape;
    ``gm(images, speed, command)`` -> (probs, mean, std, speeds) and ``gm.sample(...)`` -> actions ``[B,2]`` replay it."""

    def __init__(self, model, images, speed, command):
        if model.training:
            raise RuntimeError("GraphedMixture captures the eval-mode chain: call model.eval() first")
        self.model = model
        self.static_in = [images.clone(), speed.clone(), command.clone()]
        self.refresh()

    def refresh(self):
        """(re)capture -- after load_state_dict / parameter updates."""
        model = self.model
        with torch.no_grad():
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):                       # warm-up on the capture stream: packs weights, builds pointer tables
                    model.mixture_params(*self.static_in)
            torch.cuda.current_stream().wait_stream(side)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.static_out = model.mixture_params(*self.static_in)
        self.key = _plan_key(model)

    def __call__(self, images, speed, command):
        for dst, src in zip(self.static_in, (images, speed, command)):
            if dst.shape != src.shape:
                raise ValueError(f"GraphedMixture was captured for input shape {tuple(dst.shape)}, got {tuple(src.shape)}")
            dst.copy_(src)
        if _plan_key(self.model) != self.key:            # weights / buffers / dtype / device changed since the capture
            self.refresh()
        self.graph.replay()
        return self.static_out

    def sample(self, images, speed, command):
        probs, mean, std, _ = self(images, speed, command)
        return MixtureDistribution(probs, mean, std).sample()


class PlannedMixture:
    """The same tick WITHOUT graph capture: the launches of one eval-mode ``model.mixture_params`` call are recorded once
    (``hip.LaunchRecorder``: C-ABI function + its converted arguments, descriptors included) and re-issued per tick straight
    through ctypes -- none of the engine's Python (shape logic, descriptor filling, allocation, pointer validation) runs
    again.  The recorded run allocates from a private ``torch.cuda.MemPool`` that lives as long as the plan, so every
    recorded pointer stays valid and nothing else is handed that memory.  Same contract as :class:`GraphedMixture`: fixed
    input shapes, ``refresh()`` after a weight change, replay on the stream the plan was recorded on.  Inputs must already be
    float32 and contiguous (then the chain contains no torch kernel, only library launches)."""

    def __init__(self, model, images, speed, command):
        if model.training:
            raise RuntimeError("PlannedMixture records the eval-mode chain: call model.eval() first")
        for t in (images, speed, command):
            if t.dtype != torch.float32 or not t.is_contiguous():
                raise ValueError("PlannedMixture: inputs must be contiguous float32 tensors")
        self.model = model
        self.static_in = [images.clone(), speed.clone(), command.clone()]
        self.refresh()

    def refresh(self):
        model = self.model
        with torch.no_grad():
            for _ in range(2):                           # packs weights, builds pointer tables (cached on the parameters' versions)
                model.mixture_params(*self.static_in)
            self.pool = torch.cuda.MemPool()
            with torch.cuda.use_mem_pool(self.pool), hip.LaunchRecorder() as plan:
                self.static_out = model.mixture_params(*self.static_in)
        self.plan = plan
        self.key = _plan_key(model)

    def __call__(self, images, speed, command):
        for dst, src in zip(self.static_in, (images, speed, command)):
            if dst.shape != src.shape:
                raise ValueError(f"PlannedMixture was recorded for input shape {tuple(dst.shape)}, got {tuple(src.shape)}")
            dst.copy_(src)
        if _plan_key(self.model) != self.key:            # an eager call in between re-packed or re-allocated what the plan points to
            self.refresh()
        self.plan.replay()
        return self.static_out

    def sample(self, images, speed, command):
        probs, mean, std, _ = self(images, speed, command)
        return MixtureDistribution(probs, mean, std).sample()
