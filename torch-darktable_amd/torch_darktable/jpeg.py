"""JPEG front-end names (reference torch_darktable/jpeg.py).  The reference wraps nvjpeg; the
MI355X build has no GPU encoder (out of the kernel hot path), so `Jpeg.encode` raises."""

from .extension import extension

Jpeg = extension.Jpeg
JpegException = extension.JpegException
InputFormat = extension.JpegInputFormat
Subsampling = extension.JpegSubsampling

__all__ = ['InputFormat', 'Jpeg', 'JpegException', 'Subsampling']
