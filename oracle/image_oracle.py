"""CPU oracle (TEST INFRASTRUCTURE, not product code) for the image side of the input pipeline — SURVEY.md §8(f) rank 4.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

What it restates
----------------
reference call sites
    dexbotic/data/dataset/rgb_preprocess.py:13-28   PreprocessRGB.__call__  (pad to square -> image_processor.preprocess)
    dexbotic/data/dataset/rgb_preprocess.py:30-44   expand2square
    dexbotic/model/dexbotic_arch.py:498-529         process_images / expand2square (inference server side)
The arithmetic itself lives in two third-party dependencies that are NOT vendored under /root/reference
(pyproject.toml lists `transformers`; Pillow comes with it):
    Pillow 12.2.0  src/libImaging/Resample.c   precompute_coeffs, normalize_coeffs_8bpc,
                                               ImagingResampleHorizontal_8bpc / Vertical_8bpc  (8-bit path,
                                               PRECISION_BITS = 32 - 8 - 2, two passes with a uint8 image between)
    transformers 5.15.0  CLIPImageProcessor(Pil)  shortest-edge resize (BICUBIC) -> center crop -> x * (1/255)
                                               in float64, cast to float32 -> (x - mean) / std in float32 -> CHW
Their published algorithms are restated below in numpy; parity is pinned on tests/golden/image_t1.npz, which
oracle/gen_golden_image.py produced by running the reference's own PreprocessRGB class over the live Pillow /
transformers of this container (tests/test_image_oracle.py: uint8 stage bit-exact, float stage to 1e-6).
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2          # Resample.c: 8 bits of pixel, 2 bits of headroom for the coefficient sum
BICUBIC_SUPPORT = 2.0
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def bicubic_filter(x):
    """Resample.c bicubic_filter, a = -0.5 (Keys)"""
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size, out_size):
    """Resample.c precompute_coeffs over the full box (in0 = 0, in1 = in_size) + normalize_coeffs_8bpc.
    Returns ksize, bounds [out,2] (xmin, count) and the fixed-point taps kk [out, ksize] (int32)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = BICUBIC_SUPPORT * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)            # C (int) cast: truncation, then clipped at 0
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [bicubic_filter((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        if ww != 0.0:
            w = [v / ww for v in w]
        for x, v in enumerate(w):
            f = v * (1 << PRECISION_BITS)
            kk[xx, x] = int(-0.5 + f) if v < 0 else int(0.5 + f)
        bounds[xx] = (xmin, xmax)
    return ksize, bounds, kk


def _clip8(acc):
    return np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)     # clip8_lookups[in >> PRECISION_BITS]


def resample_axis(img, out_size, axis):
    """one pass of ImagingResample{Horizontal,Vertical}_8bpc over `axis` of a uint8 array"""
    img = np.moveaxis(img, axis, 0)
    _, bounds, kk = precompute_coeffs(img.shape[0], out_size)
    out = np.empty((out_size,) + img.shape[1:], np.uint8)
    src = img.astype(np.int64)
    for xx in range(out_size):
        xmin, n = int(bounds[xx, 0]), int(bounds[xx, 1])
        acc = np.full(img.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        acc += np.tensordot(kk[xx, :n].astype(np.int64), src[xmin:xmin + n], axes=(0, 0))
        out[xx] = _clip8(acc)
    return np.moveaxis(out, 0, axis)


def pil_resize_bicubic(img, out_h, out_w):
    """Image.resize((out_w, out_h), BICUBIC) of an RGB uint8 HWC array: horizontal pass, then vertical; a pass whose
    size does not change is skipped (ImagingResample need_horizontal / need_vertical)"""
    h, w = img.shape[:2]
    if w != out_w:
        img = resample_axis(img, out_w, 1)
    if h != out_h:
        img = resample_axis(img, out_h, 0)
    return img


def pad_color(mode, image_mean=CLIP_MEAN):
    """rgb_preprocess.py:20-23: 'zero' -> (0,0,0), otherwise int(mean * 255) per channel"""
    return (0, 0, 0) if mode == "zero" else tuple(int(x * 255) for x in image_mean)


def expand2square(img, background):
    """rgb_preprocess.py:30-44 on a uint8 HWC array"""
    h, w = img.shape[:2]
    if w == h:
        return img
    p = max(h, w)
    out = np.empty((p, p, 3), np.uint8)
    out[:] = np.asarray(background, np.uint8)
    if w > h:
        y0 = (w - h) // 2
        out[y0:y0 + h] = img
    else:
        x0 = (h - w) // 2
        out[:, x0:x0 + w] = img
    return out


def resize_output_size(h, w, shortest_edge):
    """transformers get_resize_output_image_size(default_to_square=False): short side -> shortest_edge, long side
    -> int(shortest_edge * long / short)"""
    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = shortest_edge, int(shortest_edge * long / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)      # (out_h, out_w)


def center_crop(img, ch, cw):
    h, w = img.shape[:2]
    top, left = (h - ch) // 2, (w - cw) // 2
    return img[top:top + ch, left:left + cw]


def preprocess_u8(img, aspect="pad", pad_mode="mean", size=224, crop=224, image_mean=CLIP_MEAN):
    """the integer part: [expand2square] -> shortest-edge bicubic resize -> center crop, uint8 HWC"""
    if aspect == "pad":
        img = expand2square(img, pad_color(pad_mode, image_mean))
    oh, ow = resize_output_size(img.shape[0], img.shape[1], size)
    img = pil_resize_bicubic(img, oh, ow)
    return center_crop(img, crop, crop)


def normalize(u8, image_mean=CLIP_MEAN, image_std=CLIP_STD, rescale=1 / 255):
    """x * rescale in float64 -> float32 -> (x - mean) / std in float32 -> CHW"""
    x = (u8.astype(np.float64) * rescale).astype(np.float32)
    x = (x - np.asarray(image_mean, np.float32)) / np.asarray(image_std, np.float32)
    return np.ascontiguousarray(x.transpose(2, 0, 1))


def preprocess(img, **kw):
    """PreprocessRGB.__call__ (rgb_preprocess.py:13-28) for a uint8 HWC RGB frame -> float32 [3, crop, crop]"""
    mean = kw.get("image_mean", CLIP_MEAN)
    return normalize(preprocess_u8(img, **kw), mean)


def synthetic_image(h, w, seed):
    """deterministic camera-like frame: smooth gradients + blocks + noise (uint8 HWC)"""
    rs = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.empty((h, w, 3), np.float64)
    for c in range(3):
        fx, fy, ph = rs.uniform(0.005, 0.05), rs.uniform(0.005, 0.05), rs.uniform(0, 6.28)
        img[..., c] = 127 + 90 * np.sin(fx * xx + fy * yy + ph)
    for _ in range(6):
        y0, x0 = rs.randint(0, h), rs.randint(0, w)
        y1, x1 = min(h, y0 + rs.randint(8, 1 + max(9, h // 3))), min(w, x0 + rs.randint(8, 1 + max(9, w // 3)))
        img[y0:y1, x0:x1] = rs.randint(0, 256, 3)
    img += rs.normal(0, 12, img.shape)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


# --------------------------------------------------------------------------------------------------------------------
# prompt side: dexbotic/tokenization/tokenization.py:10-31 and dexbotic/data/collator.py:16-67

def tokenizer_image_token(prompt, tokenize, bos_token_id, image_token_index=-200):
    """tokenization.py:10-31: tokenise the text either side of each '<image>', put ONE image_token_index between
    the chunks, keep a single leading BOS.  `tokenize(str) -> list[int]`."""
    chunks = [list(tokenize(c)) for c in prompt.split("<image>")]
    ids, offset = [], 0
    if chunks and chunks[0] and chunks[0][0] == bos_token_id:
        offset = 1
        ids.append(chunks[0][0])
    for n, ch in enumerate(chunks):
        if n > 0:
            ids.append(image_token_index)      # sep = [index] * (offset + 1), of which [offset:] is kept
        ids.extend(ch[offset:])
    return ids


def collate(input_ids, labels, pad_token_id, eos_token_id, model_max_length, ignore_index=-100):
    """collator.py:16-47: [eos -> -300 when pad == eos], right-pad to the longest row, truncate to
    model_max_length, mask = ids != pad, [-300 -> eos]"""
    rows = [np.asarray(r, np.int64).copy() for r in input_ids]
    same = pad_token_id == eos_token_id
    if same:
        for r in rows:
            r[r == eos_token_id] = -300
    n = max(len(r) for r in rows)
    ids = np.full((len(rows), n), pad_token_id, np.int64)
    lab = np.full((len(rows), n), ignore_index, np.int64)
    for i, (a, b) in enumerate(zip(rows, labels)):
        ids[i, :len(a)] = a
        lab[i, :len(b)] = b
    ids, lab = ids[:, :model_max_length], lab[:, :model_max_length]
    mask = ids != pad_token_id
    if same:
        ids[ids == -300] = eos_token_id
    return ids, lab, mask
