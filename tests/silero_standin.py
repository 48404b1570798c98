"""Stand-in Silero v3.1 / v4.0 archives for the tests: the generator lives in whisperjav_amd/standin_vad.py (bench.py and
scripts/ use it as a measurement-input generator too, and must not import ``tests``)."""
from whisperjav_amd.standin_vad import *        # noqa: F401,F403
from whisperjav_amd.standin_vad import VADStandIn, build, bursty_audio, get_speech_timestamps, reference_probs, save      # noqa: F401
