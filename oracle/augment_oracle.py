"""CPU ORACLE for the device-side view generation -- TEST INFRASTRUCTURE ONLY.

The reference's views are produced by torchvision 0.8.2 transforms over PIL images (datasets/EuroSAT_few_shot.py:145-170)
plus its own ImageJitter (data/additional_transforms.py:16-31).  torchvision is not installed here, but every one of those
transforms is a thin wrapper over a PIL call, restated below with PIL itself (Pillow is what the reference runs on):
Scale/Resize -> Image.resize(BILINEAR); CenterCrop -> Image.crop; RandomSizedCrop -> crop + resize(BILINEAR);
ImageEnhance.{Brightness,Contrast,Color}; flips -> Image.transpose; ToTensor -> /255 CHW; Normalize.
Given the SAME random parameters as the kernel, this is the image the reference's pipeline would produce.

Parity pin: the IMAGE arithmetic is pinned by running the real third-party code (Pillow -- present in this image and on the GPU
box -- is what executes below; torch's own float32 ops do ToTensor / Normalize).  The RANDOM DRAW SEQUENCE of the transforms lives in
torchvision (requirements.txt:20 pins torchvision==0.8.2), which is neither vendored in the reference tree nor installed here: it is
restated below from the published algorithm and anchored only on the reference's call sites -- **parity unpinned** for that half
(``tv_get_params`` / ``tv_image_view_params``)."""
import math

import numpy as np
import torch
from PIL import Image, ImageEnhance

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def _finish(img):
    """transforms.ToTensor (uint8 HWC -> float CHW, .div(255)) + transforms.Normalize (sub_(mean).div_(std)), in torch's own
    float32 arithmetic; returned HWC (the kernel writes NHWC)."""
    t = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
    mean = torch.as_tensor(MEAN, dtype=torch.float32).view(-1, 1, 1)
    std = torch.as_tensor(STD, dtype=torch.float32).view(-1, 1, 1)
    t.sub_(mean).div_(std)
    return t.permute(1, 2, 0).contiguous().numpy()


def uint8_views(src_hwc_u8, size, prm=None):
    """The PIL image (uint8 HWC) BEFORE ToTensor / Normalize: un-augmented view when ``prm`` is None."""
    if prm is None:
        return np.asarray(_noaug_pil(src_hwc_u8, size), dtype=np.uint8)
    return np.asarray(_aug_pil(src_hwc_u8, size, prm), dtype=np.uint8)


def _noaug_pil(src_hwc_u8, size):
    img = Image.fromarray(src_hwc_u8, "RGB")
    S2 = int(size * 1.15)
    img = img.resize((S2, S2), Image.BILINEAR)                      # transforms.Scale([S2, S2])
    top = int(round((S2 - size) / 2.0))
    return img.crop((top, top, top + size, top + size))             # transforms.CenterCrop(size)


def noaug_view(src_hwc_u8, size):
    return _finish(_noaug_pil(src_hwc_u8, size))


def aug_view(src_hwc_u8, size, prm):
    return _finish(_aug_pil(src_hwc_u8, size, prm))


def _aug_pil(src_hwc_u8, size, prm):
    y0, x0, h, w, rb, rc, rcol, fh, fv, _ = [np.float32(v) for v in prm]
    rb, rc, rcol = float(rb), float(rc), float(rcol)               # ImagingBlend takes a C float: the float32 value, exactly
    img = Image.fromarray(src_hwc_u8, "RGB")
    img = img.crop((int(x0), int(y0), int(x0) + int(w), int(y0) + int(h))).resize((size, size), Image.BILINEAR)
    img = ImageEnhance.Brightness(img).enhance(rb).convert("RGB")   # additional_transforms.py:27-29
    img = ImageEnhance.Contrast(img).enhance(rc).convert("RGB")
    img = ImageEnhance.Color(img).enhance(rcol).convert("RGB")
    if fh:
        img = img.transpose(Image.FLIP_LEFT_RIGHT)
    if fv:
        img = img.transpose(Image.FLIP_TOP_BOTTOM)
    return img


# ---------------------------------------------------------------------------------------------- random parameter draws
# torchvision is not vendored in the reference and not installed here; its pinned version (requirements.txt:20,
# torchvision==0.8.2) draws every transform parameter from torch's GLOBAL generator.  Published algorithm, restated:
#   RandomResizedCrop.get_params(img, scale, ratio)  [RandomSizedCrop is its deprecated alias]: up to 10 times
#       target_area = area * torch.empty(1).uniform_(scale[0], scale[1]).item()
#       aspect = torch.exp(torch.empty(1).uniform_(log(ratio[0]), log(ratio[1]))).item()
#       w = int(round(sqrt(target_area * aspect))); h = int(round(sqrt(target_area / aspect)))
#       if 0 < w <= width and 0 < h <= height: i = torch.randint(0, height - h + 1, (1,)).item(); j = torch.randint(0, width - w + 1, (1,)).item(); return
#     fallback: the largest centred crop inside the ratio bounds.
#   ImageJitter (the reference's own, data/additional_transforms.py:21-31): randtensor = torch.rand(3); r_k = alpha_k*(randtensor[k]*2.0 - 1.0) + 1
#   RandomHorizontalFlip / RandomVerticalFlip: flip if torch.rand(1) < 0.5.
# Call order for one image (datasets/EuroSAT_few_shot.py:156-170): the two un-augmented transforms draw nothing; then for each
# augmented view: get_params, ImageJitter, horizontal flip, vertical flip.

def tv_get_params(height, width, scale=(0.5, 0.9), ratio=(3.0 / 4.0, 4.0 / 3.0)):
    area = height * width
    for _ in range(10):
        target_area = area * torch.empty(1).uniform_(scale[0], scale[1]).item()
        log_ratio = torch.log(torch.tensor(ratio))
        aspect_ratio = torch.exp(torch.empty(1).uniform_(log_ratio[0], log_ratio[1])).item()
        w = int(round(math.sqrt(target_area * aspect_ratio)))
        h = int(round(math.sqrt(target_area / aspect_ratio)))
        if 0 < w <= width and 0 < h <= height:
            i = torch.randint(0, height - h + 1, size=(1,)).item()
            j = torch.randint(0, width - w + 1, size=(1,)).item()
            return i, j, h, w
    in_ratio = float(width) / float(height)
    if in_ratio < min(ratio):
        w = width
        h = int(round(w / min(ratio)))
    elif in_ratio > max(ratio):
        h = height
        w = int(round(h * max(ratio)))
    else:
        w, h = width, height
    return (height - h) // 2, (width - w) // 2, h, w


def tv_image_view_params(height, width, num_aug, jitter=(0.1, 0.1, 0.05)):
    """All draws ONE image's 2 + num_aug views consume from torch's global generator, in the reference's order
    -> float32 [2 + num_aug, 10] rows (y0, x0, h, w, r_brightness, r_contrast, r_color, flip_h, flip_v, enhance)."""
    P = np.zeros((2 + num_aug, 10), dtype=np.float32)
    P[:2, 4:7] = 1.0
    for a in range(num_aug):
        i, j, h, w = tv_get_params(height, width)
        randtensor = torch.rand(3)
        rs = [alpha * (randtensor[k] * 2.0 - 1.0) + 1 for k, alpha in enumerate(jitter)]
        fh = bool(torch.rand(1) < 0.5)
        fv = bool(torch.rand(1) < 0.5)
        P[2 + a] = [i, j, h, w, float(rs[0]), float(rs[1]), float(rs[2]), float(fh), float(fv), 1.0]
    return P
