#!/usr/bin/env python3
"""tests/golden/ref_julia_interior.npz: what the reference's own Julia screenshot (/root/reference/2.jpg, 960x512) shows INSIDE its
silhouette, as data: per 16x16 block of the image the mean linear luminance (8-bit values through gamma 2.2, the reference's display
transform, host.cpp finalize_gamma / kernel.cu:2348-2357; Rec.709 weights) and the mean linear colour of the pixels inside the
silhouette of tests/golden/ref_julia_silhouette.npz, and their number.  Run once in the build container (needs the reference's
screenshot and PIL); the GPU test test_julia_interior_luminance_against_the_references_own_screenshot reads only the fixture."""
import os
import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
z = np.load(os.path.join(HERE, "ref_julia_silhouette.npz"))
H, W = (int(v) for v in z["shape"])
mask = np.unpackbits(z["mask_bits"])[:H * W].reshape(H, W).astype(bool)
img = np.asarray(Image.open("/root/reference/2.jpg").convert("RGB")).astype(np.float64) / 255.0
assert img.shape[:2] == (H, W)
lin = img ** 2.2
lum = lin @ np.array([0.2126, 0.7152, 0.0722])
B = 16
by, bx = H // B, W // B
cnt = mask.reshape(by, B, bx, B).sum((1, 3))
sum_l = (lum * mask).reshape(by, B, bx, B).sum((1, 3))
sum_c = (lin * mask[..., None]).reshape(by, B, bx, B, 3).sum((1, 3))
with np.errstate(invalid="ignore", divide="ignore"):
    mean_l = np.where(cnt > 0, sum_l / cnt, 0.0)
    mean_c = np.where(cnt[..., None] > 0, sum_c / cnt[..., None], 0.0)
bg = np.median(img.reshape(-1, 3), axis=0)
np.savez_compressed(os.path.join(HERE, "ref_julia_interior.npz"), block=np.int64(B), count=cnt.astype(np.int32),
                    luminance=mean_l.astype(np.float32), colour=mean_c.astype(np.float32), background_8bit=(bg * 255).astype(np.float32))
print("blocks with >= 128 silhouette pixels:", int((cnt >= 128).sum()), "luminance range", float(mean_l[cnt >= 128].min()), float(mean_l[cnt >= 128].max()),
      "background", bg * 255)
