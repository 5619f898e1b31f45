from .nerf import NeRF  # noqa: F401
