"""Drop-in for the reference's ``build_model.py``: config (+) default merge, ADTModel, checkpoint load."""
import os

import torch

from adt_str_amd.config_utils import load_merged
from adt_str_amd.network import ADTModel, ADTModelConfig


def load_checkpoint_state(checkpoint_path: str):
    """``model.safetensors`` or ``pytorch_model.bin`` (possibly nested under model_state_dict / state_dict)."""
    st_path = os.path.join(checkpoint_path, "model.safetensors")
    pt_path = os.path.join(checkpoint_path, "pytorch_model.bin")
    if os.path.exists(st_path):
        from safetensors.torch import load_file
        state = load_file(st_path)
    elif os.path.exists(pt_path):
        state = torch.load(pt_path, map_location="cpu")
    else:
        raise FileNotFoundError(f"No model weights found at {checkpoint_path}")
    for key in ("model_state_dict", "state_dict"):
        if key in state:
            return state[key]
    return state


def model_config_from(cfg: dict) -> ADTModelConfig:
    section = dict(cfg.get("model", {}))
    lr = cfg.get("training", {}).get("learning_rate", 1e-4)
    section["enc_lr"] = section["dec_lr"] = lr
    section.update(cfg.get("shared", {}))
    return ADTModelConfig(**section)


def build_model(config_path: str, device: str = "cuda"):
    cfg = load_merged(config_path)
    checkpoint_path = cfg.get("inference", {}).get("checkpoint_path")
    if not checkpoint_path:
        raise ValueError("inference.checkpoint_path is required in the configuration file.")
    model = ADTModel(model_config_from(cfg))
    state = load_checkpoint_state(checkpoint_path)
    # the front-end buffers are constants: tolerate checkpoints that lack them or name them differently
    own = model.state_dict()
    missing = [k for k in own if k not in state and not k.startswith("compute_spectrogram.")]
    if missing:
        raise KeyError(f"checkpoint lacks parameters: {missing[:5]}...")
    model.load_state_dict({k: v for k, v in state.items() if k in own}, strict=False)
    model.to(torch.device(device))
    model.eval()
    return model, cfg
