"""CPU ORACLE for the device-side view generation -- TEST INFRASTRUCTURE ONLY.

The reference's views are produced by torchvision 0.8.2 transforms over PIL images (datasets/EuroSAT_few_shot.py:145-170)
plus its own ImageJitter (data/additional_transforms.py:16-31).  torchvision is not installed here, but every one of those
transforms is a thin wrapper over a PIL call, restated below with PIL itself (Pillow is what the reference runs on):
Scale/Resize -> Image.resize(BILINEAR); CenterCrop -> Image.crop; RandomSizedCrop -> crop + resize(BILINEAR);
ImageEnhance.{Brightness,Contrast,Color}; flips -> Image.transpose; ToTensor -> /255 CHW; Normalize.
Given the SAME random parameters as the kernel, this is the image the reference's pipeline would produce."""
import numpy as np
from PIL import Image, ImageEnhance

MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float64)
STD = np.array([0.229, 0.224, 0.225], dtype=np.float64)


def _finish(img):
    a = np.asarray(img, dtype=np.float64) / 255.0
    return ((a - MEAN) / STD).astype(np.float32)                   # HWC (the kernel writes NHWC)


def noaug_view(src_hwc_u8, size):
    img = Image.fromarray(src_hwc_u8, "RGB")
    S2 = int(size * 1.15)
    img = img.resize((S2, S2), Image.BILINEAR)                      # transforms.Scale([S2, S2])
    top = int(round((S2 - size) / 2.0))
    img = img.crop((top, top, top + size, top + size))              # transforms.CenterCrop(size)
    return _finish(img)


def aug_view(src_hwc_u8, size, prm):
    y0, x0, h, w, rb, rc, rcol, fh, fv, _ = [float(v) for v in prm]
    img = Image.fromarray(src_hwc_u8, "RGB")
    img = img.crop((int(x0), int(y0), int(x0) + int(w), int(y0) + int(h))).resize((size, size), Image.BILINEAR)
    img = ImageEnhance.Brightness(img).enhance(rb).convert("RGB")   # additional_transforms.py:27-29
    img = ImageEnhance.Contrast(img).enhance(rc).convert("RGB")
    img = ImageEnhance.Color(img).enhance(rcol).convert("RGB")
    if fh:
        img = img.transpose(Image.FLIP_LEFT_RIGHT)
    if fv:
        img = img.transpose(Image.FLIP_TOP_BOTTOM)
    return _finish(img)
