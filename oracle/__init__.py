"""CPU oracle for the WhisperJAV hot path (TEST INFRASTRUCTURE ONLY).

Everything under ``oracle/`` is a plain NumPy / PyTorch-CPU restatement of the
arithmetic that WhisperJAV's ``balanced`` / ``fidelity`` modes execute inside
third-party wheels (faster-whisper 1.2.1 / ctranslate2 4.7.1 / openai-whisper
20250625 / silero-vad 6.2.1 -- pins at /root/reference/uv.lock:249-4449).

Rules (enforced by tests/test_layout_rules.py):
  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
    ``cpu_baseline`` leg may import this package;
  * the product package ``whisperjav_amd`` never imports it and has no CPU
    fallback -- it raises if the HIP library is missing.

PARITY STATUS: the reference's own tests pin no numeric value at the
mel / VAD-probability / logit boundary and the upstream
wheels are not installable here, so this oracle is pinned against the
*independent* implementations that ARE importable in this container
(``transformers.WhisperFeatureExtractor`` for the log-mel formula and
``transformers...WhisperForConditionalGeneration`` for the transformer math,
see tests/test_oracle_*.py).  Search-procedure parity (CTranslate2 beam search,
patience, repetition penalty) is "parity unpinned": restated from the published
algorithm, no golden vectors from the upstream binary exist.
"""
