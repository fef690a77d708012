"""Run tools/micro/anyorder.hip on the HIP runtime the PyTorch wheel bundles (the one libia2p_hip.so runs on): import torch first, then dlopen."""
import ctypes
import os
import torch  # noqa: F401  (loads the bundled libamdhip64 first)

torch.cuda.init()
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "micro", "libanyorder.so"))
raise SystemExit(lib.anyorder_run(1))
