"""Host side of the input pipeline (§8(f) rank 4): tokenizer_image_token and the collator mirror the reference's
(dexbotic/tokenization/tokenization.py:10-31, dexbotic/data/collator.py:10-67); goldens from the reference itself
(oracle/gen_golden_image.py).  The resample tables come from the library's host routine: bit-exact with the oracle."""
import ctypes as C
import os
import types

import numpy as np
import pytest
import torch


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "image_t1.npz"), allow_pickle=False)


class Tok:
    bos_token_id, eos_token_id, pad_token_id, model_max_length = 1, 2, 0, 24

    def __call__(self, text):
        return types.SimpleNamespace(input_ids=[1] + [3 + (sum(map(ord, w)) % 997) for w in text.split()])


def test_tokenizer_image_token_matches_reference(g):
    from dexbotic_amd.tokenization.tokenization import tokenizer_image_token
    for j, prompt in enumerate(g["prompts"]):
        ids = tokenizer_image_token(str(prompt), Tok())
        assert ids == g[f"prompt{j}/ids"].tolist(), prompt
        t = tokenizer_image_token(str(prompt), Tok(), return_tensors="pt")
        assert t.dtype == torch.long and t.tolist() == ids
    with pytest.raises(ValueError):
        tokenizer_image_token("x", Tok(), return_tensors="np")


def test_collator_matches_reference(g):
    from dexbotic_amd.data.collator import DataCollatorForSupervisedDataset
    rows = [[1, 5, 2, 9, 0, 7], [1, 8, 2], list(range(1, 31))]
    labs = [[-100, -100, 2, 9, 0, 7], [-100, 8, 2], list(range(1, 31))]
    for tag, pad in (("pad0", 0), ("padeos", 2)):
        tok = Tok()
        tok.pad_token_id = pad
        inst = [{"input_ids": torch.tensor(r), "labels": torch.tensor(l), "image": torch.zeros(3, 4, 4),
                 "action": torch.zeros(7)} for r, l in zip(rows, labs)]
        b = DataCollatorForSupervisedDataset(tok)(inst)
        assert np.array_equal(b["input_ids"].numpy(), g[f"collate_{tag}/input_ids"]), tag
        assert np.array_equal(b["labels"].numpy(), g[f"collate_{tag}/labels"]), tag
        assert np.array_equal(b["attention_mask"].numpy(), g[f"collate_{tag}/attention_mask"]), tag
        assert b["images"].shape == (3, 3, 4, 4) and b["actions"].shape == (3, 7)
    ragged = [{"input_ids": torch.tensor([1]), "labels": torch.tensor([1]), "image": torch.zeros(3, 4, 4)},
              {"input_ids": torch.tensor([1]), "labels": torch.tensor([1]), "image": torch.zeros(3, 5, 5)}]
    assert isinstance(DataCollatorForSupervisedDataset(Tok())(ragged)["images"], list)


def test_resample_tables_match_oracle():
    """dxa_resample_coeffs is host code: callable without a GPU; integer tables must equal the oracle's exactly"""
    from dexbotic_amd import _lib as L
    from oracle import image_oracle as IO
    for n_in, n_out in ((640, 224), (1280, 224), (200, 224), (17, 224), (3840, 224), (500, 298), (224, 224)):
        ks, bounds, kk = IO.precompute_coeffs(n_in, n_out)
        assert L.lib.dxa_resample_ksize(n_in, n_out) == ks
        b = np.zeros((n_out, 2), np.int32)
        k = np.zeros((n_out, ks), np.int32)
        L.check(L.lib.dxa_resample_coeffs(n_in, n_out, L.FILTER_BICUBIC, b.ctypes.data_as(C.c_void_p), k.ctypes.data_as(C.c_void_p)))
        assert np.array_equal(b, bounds) and np.array_equal(k, kk), (n_in, n_out)
    assert L.lib.dxa_resample_coeffs(0, 224, L.FILTER_BICUBIC, None, None) != 0
    assert "positive" in L.last_error()


def test_conversation_templates_match_reference(g):
    from dexbotic_amd.tokenization import conversation as Cv
    want = [str(x) for x in g["conv_prompts"]]
    i = 0
    for name in ("dexbotic", "step", "llama_3"):
        for stub in (" ", None):
            c = Cv.conv_templates[name].copy()
            c.append_message(c.roles[0], "<image>\n" + "pick up the red block")
            c.append_message(c.roles[1], stub)
            assert c.get_prompt() == want[i], (name, stub)
            i += 1
        c = Cv.conv_templates[name].copy()
        c.append_message(c.roles[0], ("what is <image> this", None, "Pad"))
        c.append_message(c.roles[1], "a cube")
        c.append_message(c.roles[0], "and now?")
        assert c.get_prompt() == want[i], name
        i += 1
    assert Cv.conv_templates["dexbotic"].messages == []            # copies do not leak into the registry


class _StubModel:
    """stands in for the device model: records what the server hands over"""
    device, dtype = torch.device("cpu"), torch.float32
    config = types.SimpleNamespace(chat_template="dexbotic")

    def process_images(self, frames):
        self.sizes = [f.size for f in frames]
        return torch.zeros(len(frames), 3, 4, 4)

    def inference_action(self, ids, pix, args):
        self.ids, self.pix_shape, self.args = ids, tuple(pix.shape), args
        return [[0.5] * 7, [0.25] * 7]


def test_process_frame_wire_format(g):
    """POST /process_frame (multipart: text + PNG files) -> {"response": [[7 floats] ...]}, base_exp.py:638-653"""
    import io
    from dexbotic_amd.serve import InferenceServer, encode_png
    from dexbotic_amd.tokenization.tokenization import tokenizer_image_token
    model = _StubModel()
    srv = InferenceServer(model, Tok(), norm_stats={"min": [-1] * 7, "max": [1] * 7})
    client = srv.create_app().test_client()
    frame = np.zeros((12, 20, 3), np.uint8)
    frame[..., 0] = 200
    r = client.post("/process_frame", data={"text": "pick up the red block", "image": (io.BytesIO(encode_png(frame)), "0.png")},
                    content_type="multipart/form-data")
    assert r.status_code == 200 and r.get_json() == {"response": [[0.5] * 7, [0.25] * 7]}
    assert model.sizes == [(20, 12)] and model.pix_shape == (1, 3, 4, 4)
    assert model.args == {"cfg_scale": 1.5, "num_ddim_steps": 10, "action_norms": {"min": [-1] * 7, "max": [1] * 7}}
    want = tokenizer_image_token(str(g["conv_prompts"][0]), Tok())                 # the reference's own prompt string
    assert model.ids.tolist() == [want] and want.count(-200) == 1
    # two views: [1, views, 3, H, W]
    r = client.post("/process_frame", data={"text": "x", "image": [(io.BytesIO(encode_png(frame)), "0.png"),
                                                                    (io.BytesIO(encode_png(frame)), "1.png")]},
                    content_type="multipart/form-data")
    assert r.status_code == 200 and model.pix_shape == (1, 2, 3, 4, 4)
