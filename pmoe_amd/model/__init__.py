"""Drop-in counterpart of the reference's ``PMoE/model`` package (``from pmoe_amd.model.moe import get_model``)."""
