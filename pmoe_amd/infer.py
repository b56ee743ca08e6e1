"""Closed-loop inference helper (reference caller ``autoagents/image_agent.py:127-177``).

At B=1 the eval-mode forward is ~150 small launches and the tick time is launch latency, not kernel time.  The engine
allocates only through torch's caching allocator and (in eval mode) never synchronises with the host, so the whole
chain can be captured ONCE into a HIP graph (``torch.cuda.CUDAGraph`` is hipGraph on ROCm) and replayed per tick with
new inputs copied into the captured input buffers.  Parameters are read through the captured packed-weight buffers:
call :meth:`GraphedMixture.refresh` after loading new weights.
"""
import torch

from .model.moe import MixtureDistribution


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

    def __call__(self, images, speed, command):
        for dst, src in zip(self.static_in, (images, speed, command)):
            if dst.shape != src.shape:
                raise ValueError(f"GraphedMixture was captured for input shape {tuple(dst.shape)}, got {tuple(src.shape)}")
            dst.copy_(src)
        self.graph.replay()
        return self.static_out

    def sample(self, images, speed, command):
        probs, mean, std, _ = self(images, speed, command)
        return MixtureDistribution(probs, mean, std).sample()
