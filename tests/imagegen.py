"""Procedural test images (no image files travel to the GPU box): a smooth random blob texture
and a homography warp of it, standing in for the graf pair of BASELINE config 1."""
import numpy as np

# Ground-truth homography shipped with the Oxford graffiti set, img1 -> img4 coordinates
# (values of the reference's data file images/graf/H1to4p:1-3).
H1TO4P = np.array([[6.6378505e-01, 6.8003334e-01, -3.1230335e+01],
                   [-1.4495500e-01, 9.7128304e-01, 1.4877420e+02],
                   [4.2518504e-04, -1.3930359e-05, 1.0000000e+00]])


def texture(w, h, seed):
    """uint8 [h, w, 3] (B = G = R): sum of a few thousand Gaussian blobs."""
    rng = np.random.default_rng(seed)
    img = np.zeros((h, w), np.float64)
    n = w * h // 150
    xs, ys = rng.uniform(0, w, n), rng.uniform(0, h, n)
    rs, amp = rng.uniform(2, 9, n), rng.uniform(-1, 1, n)
    yy, xx = np.mgrid[0:h, 0:w]
    for x, y, r, a in zip(xs, ys, rs, amp):
        x0, x1 = int(max(0, x - 3 * r)), int(min(w, x + 3 * r + 1))
        y0, y1 = int(max(0, y - 3 * r)), int(min(h, y + 3 * r + 1))
        img[y0:y1, x0:x1] += a * np.exp(-((xx[y0:y1, x0:x1] - x) ** 2 + (yy[y0:y1, x0:x1] - y) ** 2) / (2 * r * r))
    img = (img - img.min()) / (img.max() - img.min())
    return np.stack([np.clip(255 * img, 0, 255).astype(np.uint8)] * 3, axis=2)


def warp(img, H):
    """View of ``img`` under homography H (source -> destination coordinates), same size."""
    from PIL import Image
    Hi = np.linalg.inv(H)
    Hi = Hi / Hi[2, 2]
    im = Image.fromarray(img)
    out = im.transform(im.size, Image.PERSPECTIVE, tuple(Hi.ravel()[:8]), Image.BICUBIC)
    return np.asarray(out, dtype=np.uint8).copy()
