"""Device gate for AMD hosts.

The reference's ``get_best_device()`` (/root/reference/whisperjav/utils/device_detector.py:112-157) detects a ROCm
device and then returns ``"cpu"`` ("ROCm support is limited ... CTranslate2 dependency"), and ``enforce_gpu_requirement``
(main.py:1621) refuses to start without ``--accept-cpu-mode``.  With the HIP drop-in that reason is gone: this module
offers the same function with the ROCm branch answering ``"cuda"`` (the device string PyTorch-ROCm uses) exactly when the
path can run there -- ``torch.version.hip`` is set, a device is visible and ``libwjhip.so`` loads with the expected ABI
-- plus ``install()``, which swaps it into the reference's module so every ``from ...device_detector import
get_best_device`` caller sees it.  INTEGRATION.md section 2e shows the one-line edit this replaces.
"""
from __future__ import annotations

import logging
from typing import Optional, Tuple

logger = logging.getLogger("whisperjav")


def hip_path_available() -> Tuple[bool, Optional[str]]:
    """(usable, gpu name): a ROCm PyTorch build, a visible AMD device and a loadable ``libwjhip.so``."""
    try:
        import torch
    except Exception as e:  # torch missing or broken: the reference treats that as "no GPU"
        logger.debug(f"torch import failed: {e}")
        return False, None
    if not getattr(torch.version, "hip", None) or not torch.cuda.is_available():
        return False, None
    try:
        name = torch.cuda.get_device_name(0)
    except RuntimeError as e:
        logger.warning(f"ROCm device initialisation failed: {e}")
        return False, None
    try:
        from . import hipbind
        hipbind.lib()
    except Exception as e:
        logger.warning(f"AMD GPU detected ({name}) but libwjhip.so is not usable: {e}")
        return False, name
    return True, name


def get_best_device(prefer_cpu: bool = False) -> str:
    """``device_detector.get_best_device`` with the AMD branch enabled.  Order as in the reference: explicit CPU
    preference, then CUDA-style devices (an NVIDIA build, or ROCm when the HIP path is usable), then MPS, then CPU."""
    if prefer_cpu:
        return "cpu"
    ok, name = hip_path_available()
    if ok:
        logger.debug(f"AMD GPU with the HIP transcription path: {name}")
        return "cuda"
    try:
        import torch
        if not getattr(torch.version, "hip", None) and torch.cuda.is_available():
            torch.cuda.get_device_name(0)
            return "cuda"
        if getattr(torch.backends, "mps", None) is not None and torch.backends.mps.is_available():
            return "mps"
    except Exception as e:
        logger.debug(f"device probe failed: {e}")
    if name:
        logger.warning(f"AMD GPU detected ({name}) but the HIP path is unavailable; using CPU mode")
    return "cpu"


def install() -> bool:
    """Replace ``whisperjav.utils.device_detector.get_best_device`` with the function above (idempotent).  Returns
    False when the ``whisperjav`` package is not importable."""
    try:
        from whisperjav.utils import device_detector  # type: ignore
    except ImportError:
        return False
    if getattr(device_detector.get_best_device, "__module__", "") != __name__:
        device_detector._reference_get_best_device = device_detector.get_best_device
        device_detector.get_best_device = get_best_device
    return True
