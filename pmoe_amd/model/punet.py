"""Drop-in parameter container for ``PMoE/model/punet.py`` (``PredictiveUnet``).

Same constructor arguments and ``state_dict`` keys (``unet.*``, ``entry_block.*``, ``pred_unet.*``); like the
reference it loads the stage-0 U-Net checkpoint ``torch.load(model_path)[model_name]`` with ``strict=False`` and
freezes it (punet.py:40-55).  Arithmetic is issued by ``pmoe_amd.engine_punet.PUNetEngine`` through the parent
``PUNetExpert`` -- the autoregressive loop of punet.py:87-120 runs there as grouped HIP launches.
"""
import torch

from . import blocks as B


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
