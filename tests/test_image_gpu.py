"""§8(f) rank 4 on the GPU: dxa_image_preprocess (through the PreprocessRGB mirror) against the golden vectors the
reference's PreprocessRGB produced with Pillow + the CLIP image processor (tests/golden/image_t1.npz) and against
oracle/image_oracle.py.  uint8 stage: bit-exact.  float stage: 1e-6 (the same float32 arithmetic)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "image_t1.npz"), allow_pickle=False)


def make(aspect, pad_mode, dtype=torch.float32):
    from dexbotic_amd.data.dataset.rgb_preprocess import ImageProcessorSpec, PreprocessRGB
    return PreprocessRGB(ImageProcessorSpec(), image_aspect_ratio=aspect, image_pad_mode=pad_mode, device=DEV, dtype=dtype)


def test_golden_frames_bit_exact(g):
    from oracle import image_oracle as IO
    for name in g["cases"]:
        h, w, seed, pad, zero = (int(v) for v in g[f"{name}/meta"])
        frame = IO.synthetic_image(h, w, seed)
        pre = make("pad" if pad else None, "zero" if zero else "mean")
        out, u8 = pre.batch(torch.from_numpy(frame)[None], want_u8=True)
        assert np.array_equal(u8[0].cpu().numpy(), g[f"{name}/u8"]), name                       # integer stage: exact
        pv = out[0].cpu().numpy()
        assert pv.shape == (3, 224, 224) and pv.dtype == np.float32
        want = IO.normalize(g[f"{name}/u8"], g["image_mean"], g["image_std"])
        assert np.max(np.abs(pv - want)) <= 1e-6 * np.max(np.abs(want)), name
        if f"{name}/pixel_values" in g.files:
            ref = g[f"{name}/pixel_values"]
            assert np.max(np.abs(pv - ref)) <= 1e-6 * np.max(np.abs(ref)), name
        # __call__ on a numpy frame gives the same single image
        assert torch.equal(pre(frame), out[0])


def test_batch_of_views_and_bf16():
    """n frames in one launch equal n single launches; bf16 output is the rounded fp32 output"""
    from oracle import image_oracle as IO
    frames = np.stack([IO.synthetic_image(480, 640, 7 + i) for i in range(5)])
    pre = make("pad", "mean")
    out, u8 = pre.batch(torch.from_numpy(frames), want_u8=True)
    for i in range(5):
        assert np.array_equal(u8[i].cpu().numpy(), IO.preprocess_u8(frames[i])), i
        assert torch.equal(out[i], pre(frames[i]))
    ob = make("pad", "mean", torch.bfloat16).batch(torch.from_numpy(frames))
    assert ob.dtype == torch.bfloat16 and torch.equal(ob, out.to(torch.bfloat16))


def test_full_size_properties():
    """size-independent checks at camera resolutions: a constant frame stays constant (taps sum to one within the
    fixed-point rounding), 224x224 input is passed through untouched, None gives zeros"""
    pre = make("pad", "mean")
    for h, w in ((1080, 1920), (2160, 3840), (224, 224)):
        frame = torch.full((1, h, w, 3), 200, dtype=torch.uint8)
        frame[..., 1] = 31
        _, u8 = pre.batch(frame, want_u8=True)
        inner = u8[0, 60:164] if h != w else u8[0]                      # rows away from the padding bands
        assert int(inner[..., 0].min()) == 200 == int(inner[..., 0].max())
        assert int(inner[..., 1].min()) == 31 == int(inner[..., 1].max())
    rs = np.random.RandomState(3)
    same = rs.randint(0, 256, (224, 224, 3)).astype(np.uint8)
    _, u8 = pre.batch(torch.from_numpy(same)[None], want_u8=True)
    assert np.array_equal(u8[0].cpu().numpy(), same)
    z = pre(None)
    assert z.shape == (3, 224, 224) and float(z.abs().max()) == 0.0


def test_process_images_on_model(golden_dir):
    """DexboticForCausalLM.process_images: PIL frames in, stacked device tensor out (dexbotic_arch.py:498-529)"""
    from PIL import Image
    from oracle import image_oracle as IO
    from .helpers import build_lm_product, load_lm_golden
    _, cfg, w = load_lm_golden(golden_dir)
    m = build_lm_product(cfg, w, "float32", DEV, train=False)
    size = m.model.mm_vision_module.image_processor.size
    s = size["shortest_edge"] if isinstance(size, dict) else size.shortest_edge
    frames = [IO.synthetic_image(120, 160, 50 + i) for i in range(2)]
    out = m.process_images([Image.fromarray(f) for f in frames])
    assert out.shape == (2, 3, s, s) and out.is_cuda
    for i, f in enumerate(frames):
        want = IO.normalize(IO.preprocess_u8(f, size=s, crop=s))
        assert np.max(np.abs(out[i].cpu().numpy() - want)) <= 1e-6 * np.max(np.abs(want))


def test_bad_arguments_fail_loudly():
    from dexbotic_amd import _lib as L
    pre = make("pad", "mean")
    with pytest.raises(L.DxaError):
        pre.batch(torch.zeros(1, 8, 8, 3))                              # not uint8
    with pytest.raises(L.DxaError):
        pre.batch(torch.zeros(1, 8, 8, 4, dtype=torch.uint8))           # not RGB


def test_process_frame_end_to_end(golden_dir):
    """PNG over the wire -> device preprocessing -> prompt -> CogACT action sampling -> JSON, against the same pieces
    called directly (oracle preprocessing, tokenizer_image_token, inference_action with the same noise draw)"""
    import io
    import types
    from dexbotic_amd.serve import InferenceServer, encode_png
    from dexbotic_amd.tokenization.tokenization import tokenizer_image_token
    from oracle import image_oracle as IO
    from .helpers import build_product, load_golden
    _, cfg, w = load_golden(golden_dir, "t1")
    m = build_product(cfg, w, "float32", DEV, train=False)
    m.eval()

    class Tok:
        bos_token_id = None

        def __call__(self, text):
            return types.SimpleNamespace(input_ids=[3 + (sum(map(ord, wd)) % (cfg.vocab_size - 3)) for wd in text.split()])

    norms = {"min": [-1.0, -2.0, -3.0, -1.0, -1.0, -1.0, 0.0], "max": [1.0, 2.0, 3.0, 1.0, 1.0, 1.0, 1.0]}
    srv = InferenceServer(m, Tok(), norm_stats=norms)
    client = srv.create_app().test_client()
    frames = [IO.synthetic_image(90, 120, 70), IO.synthetic_image(90, 120, 71)]
    s = cfg.v_image
    for views in (1, 2):
        torch.manual_seed(11)
        r = client.post("/process_frame", content_type="multipart/form-data",
                        data={"text": "pick up the red block",
                              "image": [(io.BytesIO(encode_png(f)), f"{i}.png") for i, f in enumerate(frames[:views])]})
        assert r.status_code == 200
        got = np.asarray(r.get_json()["response"])
        assert got.shape == (cfg.chunk_size, cfg.action_dim)
        pix = torch.from_numpy(np.stack([IO.normalize(IO.preprocess_u8(f, size=s, crop=s)) for f in frames[:views]])).to(DEV)
        pix = pix if views == 1 else pix[None]
        ids = tokenizer_image_token(srv.build_prompt("pick up the red block"), Tok(), return_tensors="pt")[None].to(DEV)
        torch.manual_seed(11)
        want = np.asarray(m.inference_action(ids, pix, {"cfg_scale": 1.5, "num_ddim_steps": 10, "action_norms": norms}))
        assert np.max(np.abs(got - want)) <= 1e-5 * np.max(np.abs(want)), views


def test_random_and_extreme_frame_sizes():
    """sizes the goldens do not hold: extreme aspect ratios, up-scaling in one axis and down-scaling in the other,
    1-pixel sides — device result against the oracle (itself pinned to Pillow), uint8 stage exact"""
    from oracle import image_oracle as IO
    rs = np.random.RandomState(9)
    sizes = [(int(rs.randint(3, 700)), int(rs.randint(3, 900))) for _ in range(6)] + [(1, 50), (50, 1), (224, 3), (2000, 37), (37, 2000)]
    for h, w in sizes:
        frame = rs.randint(0, 256, (h, w, 3)).astype(np.uint8)
        for aspect in ("pad", None):
            pre = make(aspect, "mean")
            out, u8 = pre.batch(torch.from_numpy(frame)[None], want_u8=True)
            want = IO.preprocess_u8(frame, aspect=aspect)
            assert np.array_equal(u8[0].cpu().numpy(), want), (h, w, aspect)
            ref = IO.normalize(want)
            assert np.max(np.abs(out[0].cpu().numpy() - ref)) <= 1e-6 * np.max(np.abs(ref)), (h, w, aspect)
