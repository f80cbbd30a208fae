"""clsr_amd: the CLSR training / scoring step as hand-written gfx950 HIP kernels behind the reference's Python API."""
import os

# The step runs on four HIP streams (main, long-term-attention branch, small auxiliary kernels, weight-gradient
# partial sums; clsr_amd/net.py).  ROCm maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4, shared with
# the runtime's own streams): with the default two of the four alias and the overlap is lost (4.80 vs 4.63 ms/step,
# DESIGN.md section 3).  Read by the HIP runtime when it initialises, i.e. at the first device call.
if "GPU_MAX_HW_QUEUES" not in os.environ:
    import sys as _sys

    _torch = _sys.modules.get("torch")
    if _torch is not None and getattr(_torch, "cuda", None) is not None and _torch.cuda.is_initialized():
        import warnings as _warnings

        _warnings.warn("clsr_amd imported after the HIP runtime was initialised: GPU_MAX_HW_QUEUES=8 can no longer "
                       "take effect (set it in the environment, or import clsr_amd before the first device call); "
                       "the training step runs ~4 % slower with the default of 4 hardware queues")
    os.environ["GPU_MAX_HW_QUEUES"] = "8"
