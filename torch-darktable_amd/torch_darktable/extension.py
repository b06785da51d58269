"""Binds the name `extension` to the MI355X kernel module (reference torch_darktable/extension.py:3)."""

from . import torch_darktable_extension as extension

__all__ = ['extension']
