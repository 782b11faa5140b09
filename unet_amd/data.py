"""Input side of the training loop without image codecs: `.npy` datasets, a synthetic generator, and the
statistics the byte-input feed (`NeuralNet.set_input_normalization` / `train_u8` / `infer_u8`) needs.

The reference's driver decodes tif files, stacks them, divides by 255 and z-scores per channel on the host
(main.py:20-58, 85-98) and hands float32 [N,C,H,W] to `train`.  Here the images stay uint8 [N,H,W,C] -- the
layout image files decode to -- and the normalisation runs on the GPU, so a batch crosses PCIe as bytes.
"""
import numpy as np


def channel_stats(images_u8, chunk=64):
    """Per-channel mean and standard deviation of images/255 over (N, H, W), accumulated in float64 in
    chunks of `chunk` images (main.py:89-91 computes the same numbers from a float32 copy of the whole set)."""
    images_u8 = np.asarray(images_u8)
    if images_u8.dtype != np.uint8 or images_u8.ndim != 4:
        raise ValueError("images must be uint8 [N, H, W, C]")
    c = images_u8.shape[3]
    s1, s2, count = np.zeros(c), np.zeros(c), 0
    for i in range(0, images_u8.shape[0], chunk):
        x = images_u8[i:i + chunk].astype(np.float64) / 255.0
        s1 += x.sum(axis=(0, 1, 2))
        s2 += (x * x).sum(axis=(0, 1, 2))
        count += x.shape[0] * x.shape[1] * x.shape[2]
    mean = s1 / count
    var = np.maximum(s2 / count - mean * mean, 0.0)
    return mean.astype(np.float32), np.sqrt(var).astype(np.float32)


def normalize_host(images_u8, mean, std):
    """What main.py:85-92 hands to `train`: float32 [N,C,H,W], (x/255 - mean) / (std + 1e-7)."""
    x = np.transpose(np.asarray(images_u8), (0, 3, 1, 2)).astype(np.float32) / np.float32(255.0)
    m = np.asarray(mean, np.float32).reshape(1, -1, 1, 1)
    s = np.asarray(std, np.float32).reshape(1, -1, 1, 1)
    return np.ascontiguousarray((x - m) / (s + np.float32(1e-7)))


def masks_host(masks_u8):
    """main.py:93-98: float32 masks, divided by 255 when stored as 0/255."""
    m = np.asarray(masks_u8).astype(np.float32)
    if m.size and m.max() > 1.0:
        m = m / np.float32(255.0)
    return np.ascontiguousarray(m)


def load_npy(images_path, masks_path, mmap=True):
    """uint8 images [N,H,W,C] and uint8 masks [N,H,W] from two .npy files (memory-mapped by default, so a
    dataset larger than host memory streams batch by batch).  Pickled arrays are refused."""
    mode = "r" if mmap else None
    images = np.load(images_path, mmap_mode=mode, allow_pickle=False)
    masks = np.load(masks_path, mmap_mode=mode, allow_pickle=False)
    if images.dtype != np.uint8 or images.ndim != 4:
        raise ValueError(f"{images_path}: expected uint8 [N, H, W, C], got {images.dtype} {images.shape}")
    if masks.dtype != np.uint8 or masks.ndim != 3:
        raise ValueError(f"{masks_path}: expected uint8 [N, H, W], got {masks.dtype} {masks.shape}")
    if masks.shape != images.shape[:3]:
        raise ValueError(f"mask shape {masks.shape} does not match image shape {images.shape}")
    return images, masks


def synthetic_dataset(n, h=256, w=256, c=3, seed=0, mask_value=255):
    """Blob-on-noise images with matching elliptical masks (about 3 % foreground, like the tumour masks the
    reference trains on): uint8 images [n,h,w,c], uint8 masks [n,h,w] holding 0 / mask_value."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    images = np.empty((n, h, w, c), np.uint8)
    masks = np.zeros((n, h, w), np.uint8)
    for i in range(n):
        cy, cx = rng.uniform(0.2, 0.8) * h, rng.uniform(0.2, 0.8) * w
        ry, rx = rng.uniform(0.05, 0.14) * h, rng.uniform(0.05, 0.14) * w
        inside = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0
        base = rng.normal(90.0, 25.0, size=(h, w, c))
        base[inside] += rng.uniform(50.0, 90.0)
        images[i] = np.clip(base, 0, 255).astype(np.uint8)
        masks[i][inside] = mask_value
    return images, masks
