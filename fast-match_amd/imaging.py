"""Minimal host-side image helpers used only to feed SIFT (out of the accelerated path).

The reference's ``imaging.py`` (PIL two-pass thumbnail ``imaging.py:49-55``, cv2
``open_img`` ``imaging.py:80-85``) is host image I/O and stays on the CPU; this shim
offers the three calls the hot path's callers need (``get_thumbnail``, ``get_size``,
``open_img``) on top of Pillow, and accepts ``cache.Feature_Image`` objects unchanged."""
import numpy as np


def _pil():
    try:
        from PIL import Image
    except ImportError:
        raise Exception("Pillow is required to load or resize images")
    return Image


def get_size(img):
    """(width, height) of an image array or Feature_Image."""
    return (int(img.shape[1]), int(img.shape[0]))


def open_img(path, max_size=-1):
    Image = _pil()
    if isinstance(path, bytes):
        path = path.decode()
    im = Image.open(path).convert("RGB")
    if max_size is not None and max_size > 0 and max(im.size) > max_size:
        s = float(max_size) / max(im.size)
        im = im.resize((max(1, int(im.size[0] * s)), max(1, int(im.size[1] * s))), Image.LANCZOS)
    return np.asarray(im, dtype=np.uint8)[:, :, ::-1].copy()      # BGR like cv2.imread


def get_thumbnail(img, size=(400, 400)):
    """Aspect-preserving thumbnail no larger than ``size`` (w, h) as uint8[H, W, 3]."""
    Image = _pil()
    if isinstance(img, (str, bytes)):
        img = open_img(img)
    im = Image.fromarray(np.ascontiguousarray(img[:, :, ::-1]))
    w, h = im.size
    scale = min(float(size[0]) / w, float(size[1]) / h, 1.0)
    tw, th = max(1, int(w * scale)), max(1, int(h * scale))
    # two passes like the reference: a cheap 2x oversize reduction, then antialias
    if w > 2 * tw and h > 2 * th:
        im = im.resize((2 * tw, 2 * th), Image.NEAREST)
    im = im.resize((tw, th), Image.LANCZOS)
    return np.asarray(im, dtype=np.uint8)[:, :, ::-1].copy()
