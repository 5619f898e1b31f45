"""Checkpoint loading for the renderer (API of nerfmatch/nerf_evaluator.py:119-156).

`load_nerf_render_from_ckpt(ckpt_path, device, stop_layer)` reads a Lightning checkpoint written by the reference
(`hyper_parameters` = nested Namespace, `state_dict` with a `model.` prefix) and returns a NerfRenderer whose
weights are packed for the HIP kernels.  torch >= 2.6 defaults to weights_only=True, which rejects these pickles,
hence weights_only=False."""
from argparse import Namespace

import torch

from .nerf.renderer import NerfRenderer


def local_device():
    """The GPU of this process: LOCAL_RANK when a launcher set it (one process per GPU; the device is made current, because
    the kernels run on the current device's stream), else the current device.  The reference pins cuda:0
    (nerf_evaluator.py:153) -- it never runs more than one evaluation process."""
    import os

    if not torch.cuda.is_available():
        return torch.device("cpu")
    lr = os.environ.get("LOCAL_RANK")
    if lr is not None and int(lr) < torch.cuda.device_count():
        torch.cuda.set_device(int(lr))
    return torch.device("cuda", torch.cuda.current_device())


class GenericModelEvaluator(torch.nn.Module):
    """reference: nerf_evaluator.py:149-156 (device selection, grad globally off)."""

    def __init__(self, config):
        super().__init__()
        self.device = local_device()
        torch.set_grad_enabled(False)
        self.config = config


def load_nerf_render_from_ckpt(ckpt_path, device, stop_layer=-1, unnorm_scene=None):
    ckpt = torch.load(ckpt_path, map_location="cpu", weights_only=False)
    state = ckpt["state_dict"]
    vocab = state["model.embedding_a.weight"].shape[0] if "model.embedding_a.weight" in state else -1
    hp = ckpt["hyper_parameters"]
    config = hp if isinstance(hp, Namespace) else Namespace(**hp)
    render = NerfRenderer(config, num_frames=vocab, training=False, stop_layer=stop_layer)
    new_state = {k[len("model."):]: v for k, v in state.items() if k.startswith("model.")}
    missing = render.load_state_dict(new_state, strict=False)
    if [k for k in missing.missing_keys if "scales" not in k]:
        raise RuntimeError(f"checkpoint lacks NeRF weights: {missing.missing_keys}")
    render.to(device).eval()
    # The reference recomputes the scene normalisation from the training transforms json
    # (nerf_evaluator.py:99-116): a one-off CPU/JSON step outside the hot path; pass it in (or store it in the ckpt).
    render.unnorm_scene = unnorm_scene if unnorm_scene is not None else ckpt.get("unnorm_scene")
    return render


def save_nerf_ckpt(path, config, state_dict, unnorm_scene=None, epoch=0, global_step=0):
    """Writes a checkpoint with the reference's Lightning layout (used by tests / synthetic benchmarks)."""
    ckpt = dict(state_dict={f"model.{k}": v for k, v in state_dict.items()}, hyper_parameters=vars(config), epoch=epoch,
                global_step=global_step)
    if unnorm_scene is not None:
        ckpt["unnorm_scene"] = unnorm_scene
    torch.save(ckpt, path)
