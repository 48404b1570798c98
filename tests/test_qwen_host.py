"""Host side of the Qwen3 slice (no GPU): blob packing matches the header's tensor order, the TextGenerator adapter has the
protocol's surface and refuses to run without its plug-ins."""
import re
import os

import numpy as np
import pytest

from whisperjav_amd import hipbind, qwen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_blob_layout_matches_the_header():
    text = open(os.path.join(ROOT, "include", "wjhip.h")).read()
    layer = re.search(r"enum \{ WJ_QL_LN1_W = 0,([^}]*)WJ_QL_N \}", text).group(1)
    names = ["LN1_W"] + [n.strip().replace("WJ_QL_", "") for n in layer.split(",") if n.strip()]
    assert tuple(names) == qwen.LAYER_TENSORS
    d = qwen.Qwen3Dims(hidden=64, n_layer=2, n_head=2, n_kv_head=1, head_dim=128, ffn=96, vocab=50)
    w = qwen.synth_weights(d)
    tensors = qwen.engine_tensors(d, w)
    assert len(tensors) == 2 + d.n_layer * len(qwen.LAYER_TENSORS)
    assert tensors[2 + 1][1].shape == ((d.n_head + 2 * d.n_kv_head) * d.head_dim, d.hidden)      # fused q | k | v rows
    assert tensors[2 + 6][1].shape == (2 * d.ffn, d.hidden)                                        # gate rows, then up rows
    blob, offsets = qwen.pack_blob(d, w, "float16")
    assert len(offsets) == len(tensors) and all(o % 256 == 0 for o in offsets)
    import ctypes
    import torch
    emb = blob[int(offsets[0]): int(offsets[0]) + d.vocab * d.hidden * 2].view(torch.float16).float().numpy()
    assert np.allclose(emb.reshape(d.vocab, d.hidden), w["model.language_model.embed_tokens.weight"], atol=2e-3)
    assert ctypes.sizeof(qwen.Qwen3DimsC) == 7 * 4 + 2 * 4          # wj_qwen_dims


def test_text_generator_surface_and_refusal(tmp_path):
    d = qwen.Qwen3Dims(hidden=64, n_layer=1, n_head=1, n_kv_head=1, head_dim=128, ffn=64, vocab=32)
    gen = qwen.HipQwenTextGenerator(d, qwen.synth_weights(d))
    for name in ("generate", "generate_batch", "load", "unload", "cleanup"):      # subtitle_pipeline/protocols.py:60-110
        assert callable(getattr(gen, name))
    with pytest.raises(hipbind.WjError, match="audio_embedder"):
        gen.generate(tmp_path / "x.wav")
    gen.cleanup(); gen.cleanup()
