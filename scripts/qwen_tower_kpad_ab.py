"""A/B of the audio tower's convolution stem with K = 9 C padded to a multiple of 64 (wj_tune qwen_conv_kpad): published tower
geometry, seeded weights, 512 clips of 4 s resident in HBM; prints ms per encode for the switch on and off."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from whisperjav_amd import hipbind, qwen  # noqa: E402

ad = qwen.Qwen3AudioDims()
w = qwen.synth_audio_weights(ad, seed=4)
rng = np.random.default_rng(1)
clips = [torch.from_numpy((rng.standard_normal(64000) * 0.1).astype(np.float32)).cuda() for _ in range(512)]
out = {}
for mode in (1, 0):
    hipbind.tune("qwen_conv_kpad", mode)
    tower = qwen.HipQwenAudioTower(ad, w, dtype="float16", max_seconds=1024)
    tower.encode(clips)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        e = tower.encode(clips)
    torch.cuda.synchronize()
    out[f"kpad_{mode}_ms"] = round((time.perf_counter() - t0) * 500.0, 2)
    out[f"kpad_{mode}_sum"] = float(sum(float(x.float().abs().mean()) for x in e[:8]))
    tower.close()
    print(json.dumps(out), flush=True)
hipbind.tune("qwen_conv_kpad", 1)
