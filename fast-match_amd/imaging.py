"""Host-side image helpers that feed the feature extractor (out of the accelerated path).

Mirrors the three calls of the reference's ``imaging.py`` that the hot path's callers use --
``get_thumbnail`` (reference ``imaging.py:67-69`` -> ``scale_pil_antialias``, ``:49-55``),
``get_size`` (``:71-72``) and ``open_img`` (``:80-85``) -- on Pillow alone (the reference also
needs cv2 for ``open_img``).  Same argument meaning: ``path`` is a file name or an image
array, ``size`` a (width, height) box.  Differences forced by the environment: Pillow >= 10
has no ``Image.ANTIALIAS`` (``LANCZOS`` is the same filter) and no Python-2 ``basestring``;
``open_img`` resizes with Pillow's BOX filter where the reference uses cv2 ``INTER_AREA``.
``cache.Feature_Image`` objects pass through ``get_size`` unchanged."""
import numpy as np


def _pil():
    try:
        from PIL import Image
    except ImportError:
        raise Exception("Pillow is required to load or resize images")
    return Image


def _is_path(data):
    return isinstance(data, (str, bytes))


def _open_pil(data):
    """PIL image from a path or an array (reference scale_pil.open / from_array, imaging.py:41-42)."""
    Image = _pil()
    if _is_path(data):
        return Image.open(data.decode() if isinstance(data, bytes) else data)
    return Image.fromarray(np.ascontiguousarray(data))


def resize_rule(width, height, size=(200, 200)):
    """Target size of the reference's ``scale.resize`` (imaging.py:28-36): the longer side
    takes the box's extent on that axis, the other follows the aspect ratio (truncated)."""
    if width > height:
        w = size[0]
        h = int((w / float(width)) * height)
    else:
        h = size[1]
        w = int((h / float(height)) * width)
    return (w, h)


def get_thumbnail(path, size=(200, 200)):
    """ Get thumbnail with PIL: two ``Image.thumbnail`` passes, first to twice the target size
    with the default filter, then to the target with the antialias (Lanczos) filter
    (reference imaging.py:49-55).  Returns uint8 [h, w, channels]. """
    Image = _pil()
    img = _open_pil(path)
    new_size = resize_rule(img.size[0], img.size[1], size)
    img.thumbnail(tuple(i * 2 for i in new_size))
    img.thumbnail(new_size, Image.LANCZOS)
    return np.array(img, dtype=np.uint8)


def get_size(path):
    """(width, height) of an image file, an image array or a Feature_Image."""
    if _is_path(path):
        return _open_pil(path).size
    return (int(path.shape[1]), int(path.shape[0]))


def open_img(path, size=None):
    """ Image as uint8 [H, W, 3] in cv2.imread's BGR channel order; ``size`` None or -1 keeps
    the resolution, a (width, height) box -- or one number for a square box -- rescales by the rule above (reference imaging.py:80-85). """
    Image = _pil()
    img = _open_pil(path).convert("RGB")
    if not (size is None or (np.isscalar(size) and size <= 0)):
        if np.isscalar(size):                      # Metric_Cache passes options["max_size"], an int (cache.pyx:158, 266)
            size = (size, size)
        img = img.resize(resize_rule(img.size[0], img.size[1], size), Image.BOX)
    return np.asarray(img, dtype=np.uint8)[:, :, ::-1].copy()
