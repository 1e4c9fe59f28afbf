from .unet_model import UNet  # noqa: F401  (same import surface as the reference's unet/__init__.py:1)
