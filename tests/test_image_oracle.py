"""Pins oracle/image_oracle.py (numpy restatement of Pillow's 8-bit bicubic resample + the CLIP image processor and of
the reference's tokenizer_image_token / collator) against tests/golden/image_t1.npz, which oracle/gen_golden_image.py
produced with the reference's own PreprocessRGB / tokenizer_image_token / DataCollatorForSupervisedDataset."""
import os
import types

import numpy as np
import pytest

from oracle import image_oracle as IO


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "image_t1.npz"), allow_pickle=False)


def case_args(meta):
    h, w, seed, pad, zero = (int(v) for v in meta)
    return IO.synthetic_image(h, w, seed), dict(aspect="pad" if pad else None, pad_mode="zero" if zero else "mean")


def test_uint8_stage_bit_exact(g):
    for name in g["cases"]:
        frame, kw = case_args(g[f"{name}/meta"])
        u8 = IO.preprocess_u8(frame, **kw)
        assert u8.dtype == np.uint8 and np.array_equal(u8, g[f"{name}/u8"]), name


def test_float_stage(g):
    for name in g["cases"]:
        pv = IO.normalize(g[f"{name}/u8"], g["image_mean"], g["image_std"])
        assert abs(pv.astype(np.float64).sum() - float(g[f"{name}/pv_sum"])) < 1e-6 * float(g[f"{name}/pv_abs"]), name
        if f"{name}/pixel_values" in g.files:
            want = g[f"{name}/pixel_values"]
            assert np.max(np.abs(pv - want)) <= 1e-6 * np.max(np.abs(want)), name


def test_coefficients_are_a_partition_of_unity():
    for n_in, n_out in ((640, 224), (1280, 224), (200, 224), (224, 224), (17, 224)):
        ksize, bounds, kk = IO.precompute_coeffs(n_in, n_out)
        s = kk.sum(1)
        assert np.all(np.abs(s - (1 << IO.PRECISION_BITS)) <= ksize)          # rounding of <= ksize taps
        assert np.all(bounds[:, 0] >= 0) and np.all(bounds[:, 0] + bounds[:, 1] <= n_in)


class _Tok:
    bos_token_id = 1

    def __call__(self, text):
        return types.SimpleNamespace(input_ids=[1] + [3 + (sum(map(ord, w)) % 997) for w in text.split()])


def test_tokenizer_image_token(g):
    tok = _Tok()
    for j, prompt in enumerate(g["prompts"]):
        ids = IO.tokenizer_image_token(str(prompt), lambda s: tok(s).input_ids, tok.bos_token_id)
        assert np.array_equal(np.asarray(ids, np.int64), g[f"prompt{j}/ids"]), prompt


def test_collate(g):
    rows = [[1, 5, 2, 9, 0, 7], [1, 8, 2], list(range(1, 31))]
    labs = [[-100, -100, 2, 9, 0, 7], [-100, 8, 2], list(range(1, 31))]
    for tag, pad in (("pad0", 0), ("padeos", 2)):
        ids, lab, mask = IO.collate(rows, labs, pad, 2, 24)
        assert np.array_equal(ids, g[f"collate_{tag}/input_ids"]), tag
        assert np.array_equal(lab, g[f"collate_{tag}/labels"]), tag
        assert np.array_equal(mask, g[f"collate_{tag}/attention_mask"]), tag


def test_oracle_against_live_pillow_on_random_sizes():
    """beyond the committed goldens: the restatement against whatever Pillow is installed (same package the reference
    resizes with), random frame sizes incl. extreme aspect ratios and up-scaling; skipped where Pillow is absent"""
    Image = pytest.importorskip("PIL.Image")
    rs = np.random.RandomState(5)
    sizes = [(rs.randint(3, 700), rs.randint(3, 900)) for _ in range(12)] + [(1, 50), (50, 1), (224, 3), (2000, 37)]
    for n, (h, w) in enumerate(sizes):
        frame = rs.randint(0, 256, (h, w, 3)).astype(np.uint8)
        for aspect in ("pad", None):
            img = IO.expand2square(frame, IO.pad_color("mean")) if aspect == "pad" else frame
            oh, ow = IO.resize_output_size(img.shape[0], img.shape[1], 224)
            want = np.asarray(Image.fromarray(img).resize((ow, oh), resample=Image.BICUBIC))
            got = IO.pil_resize_bicubic(img, oh, ow)
            assert np.array_equal(got, want), (n, h, w, aspect)
