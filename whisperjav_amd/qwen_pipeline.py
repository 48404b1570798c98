"""``--mode qwen`` (BASELINE cfg5) over the reference's own orchestrator, with the per-scene loops pooled.

``DecoupledSubtitlePipeline`` (reference whisperjav/modules/subtitle_pipeline/orchestrator.py:61-1157) frames every scene,
then asks the generator for the text of ONE scene's frames at a time (:461-492) and the aligner for one scene's words at a
time (:632-638).  On one MI355X the decoder streams its 3.4 GB of weights per generated position whatever the number of rows,
and 288 GB of HBM hold the K/V caches of thousands of clips -- the whole recording is one batch (Qwen3-ASR section:
5750x real-time at 1800 clips per step against 1359x at 64).  ``hip_decoupled_pipeline_class()`` returns a subclass that
announces every frame of every scene to the adapters' ``prime`` seam before the inherited step runs; the inherited loops, their
error handling (batch -> per-frame fall-back), artifacts, VRAM lifecycle and everything after them (sentinel, reconstruction,
hardening) are the reference's code, unchanged.  Adapters without ``prime`` (the reference's own) are driven exactly as before.

The subclass is built on demand because the reference package is an optional import of this package (it is the HOST of the
plug-in, not a dependency): ``QwenPipeline._build_subtitle_pipeline`` (pipelines/qwen_pipeline.py:389-530) returns
``DecoupledSubtitlePipeline(...)`` -- a maintainer replaces that name with ``hip_decoupled_pipeline_class()`` (INTEGRATION.md f).
"""
from __future__ import annotations

import logging
from typing import Any, List, Optional

logger = logging.getLogger("whisperjav_amd")


def hip_decoupled_pipeline_class():
    from whisperjav.modules.subtitle_pipeline.orchestrator import DecoupledSubtitlePipeline

    class HipDecoupledSubtitlePipeline(DecoupledSubtitlePipeline):
        """The reference orchestrator with generation and alignment pooled over all scenes."""

        def _step2_4_generate_and_clean(self, scene_frames, frame_audio_paths, scene_durations):
            prime = getattr(self.generator, "prime", None)
            if prime is not None:
                paths, durations = [], []
                for frames, audio_paths in zip(scene_frames, frame_audio_paths):
                    for frame, path in zip(frames, audio_paths):
                        if frame.text is None:                       # frames that carry text are not generated (:470-476)
                            paths.append(path)
                            durations.append(frame.duration)
                if paths:
                    try:
                        prime(paths, language=self.language, contexts=[self.context] * len(paths) if self.context else None,
                              audio_durations=durations)
                    except Exception:
                        # the inherited loop then generates scene by scene, with its own per-frame fall-back
                        logger.warning("pooled generation failed; falling back to the per-scene calls", exc_info=True)
            return super()._step2_4_generate_and_clean(scene_frames, frame_audio_paths, scene_durations)

        def _step5_7_align(self, scene_frames, frame_audio_paths, scene_texts, scene_durations):
            prime = getattr(self.aligner, "prime", None) if self.aligner is not None else None
            if prime is not None:
                paths: List[Any] = []
                texts: List[str] = []
                durations: List[float] = []
                for frames, audio_paths, frame_texts in zip(scene_frames, frame_audio_paths, scene_texts):
                    for frame, path, text in zip(frames, audio_paths, frame_texts):
                        if text.strip():                              # empty frames are not aligned (:613-618)
                            paths.append(path); texts.append(text); durations.append(frame.duration)
                if paths:
                    try:
                        prime(paths, texts, language=self.language, audio_durations=durations)
                    except Exception:
                        logger.warning("pooled alignment failed; falling back to the per-scene calls", exc_info=True)
            return super()._step5_7_align(scene_frames, frame_audio_paths, scene_texts, scene_durations)

    return HipDecoupledSubtitlePipeline
