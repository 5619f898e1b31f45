from .config import dict2namespace, namespace2dict, merge_configs, update_configs, load_yaml_config  # noqa: F401


def data_to_device(data, device):
    """In-place move of every tensor value of a batch dict (reference: nerfmatch/utils/__init__.py:16-19)."""
    import torch

    for k, v in data.items():
        if isinstance(v, torch.Tensor):
            data[k] = v.to(device)
