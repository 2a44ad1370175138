"""GEMM solution selection for the dense projections (PyTorch TunableOp over rocBLAS / hipBLASLt).

The [E,128]x[128,128|256] products of the layer are far from hipBLASLt's default heuristics'
sweet spot (76 TF/s by default vs 107-126 TF/s with the best available solution, measured on
MI355X).  ``tuned/tunableop_gfx950.csv`` holds the solutions picked by a tuning run of bench.py's
workload on this image (``PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 python bench.py``);
``enable_tuned_gemms()`` loads it with tuning switched off, so shapes not in the file (or a file
whose validators -- torch / ROCm / hipBLASLt versions, gfx arch -- do not match) simply fall back
to the default heuristic.  Results are bit-wise those of the selected library kernels; fp32 throughout.
"""
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_FILE = os.path.join(_HERE, "tuned", "tunableop_gfx950.csv")


TUNE_OUT = None     # development (a module attribute): a file name -> tune the shapes this run meets and write them there


def enable_tuned_gemms(path=None, allow_tuning=False):
    """Returns True if the tuned-solution file was loaded."""
    import torch
    if not torch.cuda.is_available():
        return False
    import torch.cuda.tunable as tn
    path = path or DEFAULT_FILE
    tune_out = TUNE_OUT                             # development: tune the shapes this run meets and write them to this file
    if tune_out:
        allow_tuning = True
        tn.set_filename(tune_out)
    tn.enable(True)
    tn.tuning_enable(bool(allow_tuning))
    try:
        tn.write_file_on_exit(bool(allow_tuning))
    except Exception:
        pass
    if os.path.exists(path):
        try:
            return bool(tn.read_file(path))
        except Exception:
            return False
    return False
