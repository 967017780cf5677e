"""TEST INFRASTRUCTURE -- writes tests/golden/jpeg_cases.npz: JPEG files (bytes) written by Pillow and the pixels Pillow's own decoder
(`Image.open(...).convert("RGB")`, the reference's call: anakin/datasets/ho3d.py:228-231) returns for them.  Data only: the files and the
expected arrays.  Run in the build container (Pillow 12.2.0 / libjpeg-turbo 3.1.4.1); the reference pins Pillow==8.0.1 (requirements.txt:94),
same decoder defaults (JDCT_ISLOW, fancy up-sampling)."""
import io
import os

import numpy as np
from PIL import Image, ImageFile

ImageFile.MAXBLOCK = 1 << 24
HERE = os.path.dirname(os.path.abspath(__file__))


def picture(w, h, kind, rng):
    y, x = np.mgrid[0:h, 0:w]
    base = np.stack([128 + 100 * np.sin(x / 9.0 + y / 17.0), 128 + 90 * np.cos(x / 5.0) * np.sin(y / 7.0), (x * 3 + y * 2) % 256], -1)
    if kind == "noise":
        base = base + rng.normal(0, 40, (h, w, 3))
    if kind == "hard":
        base = rng.integers(0, 256, (h, w, 3))
    return np.clip(base, 0, 255).astype(np.uint8)


CASES = [  # (w, h, subsampling, quality, kind, extra save arguments)
    (64, 48, 0, 90, "noise", {}), (64, 48, 1, 75, "noise", {}), (64, 48, 2, 92, "noise", {}),
    (70, 50, 2, 85, "smooth", {"optimize": True}), (33, 17, 1, 95, "hard", {}), (17, 33, 2, 100, "hard", {}),
    (1, 1, 2, 90, "noise", {}), (2, 2, 2, 90, "noise", {}), (5, 3, 1, 90, "noise", {}), (3, 5, 2, 60, "noise", {}),
    (127, 129, 2, 30, "noise", {"restart_marker_blocks": 3}), (96, 80, 1, 88, "noise", {"restart_marker_rows": 1}),
    (160, 120, 2, 92, "noise", {}), (160, 120, 0, 92, "smooth", {"restart_marker_blocks": 1}),
    (64, 48, None, 85, "grey", {}), (7, 9, None, 50, "grey", {"optimize": True}),
]


def main():
    rng = np.random.default_rng(7)
    out = {"n": np.int64(len(CASES))}
    for i, (w, h, sub, q, kind, extra) in enumerate(CASES):
        b = io.BytesIO()
        if kind == "grey":
            Image.fromarray(picture(w, h, "noise", rng)[..., 0]).save(b, "JPEG", quality=q, **extra)
        else:
            Image.fromarray(picture(w, h, kind, rng)).save(b, "JPEG", quality=q, subsampling=sub, **extra)
        data = b.getvalue()
        out[f"file{i}"] = np.frombuffer(data, np.uint8)
        out[f"rgb{i}"] = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
    np.savez_compressed(os.path.join(HERE, "..", "tests", "golden", "jpeg_cases.npz"), **out)


if __name__ == "__main__":
    main()
