import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from whisperjav_amd import vad_graph, synth
from whisperjav_amd.standin_vad import build
a = torch.from_numpy(synth.speech_like(600.0, seed=1, noisy=True)).cuda()
clips = [a[i*160000:(i+1)*160000+777] for i in range(59)]
sc = vad_graph.HipGraphVadScorer(build("v4", seed=7))
for i, t in enumerate(sc.program.listing): print(i, t, file=sys.stderr)
sc.scores(clips)
os.environ["X"]="1"
sc.scores(clips * 12)
