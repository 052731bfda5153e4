"""Load the host-interpreted build of the kernel sources (tests/hipemu) and inject it into the binding layer.
TEST INFRASTRUCTURE: lets the CPU-only build container execute the real csrc/*.hip code paths."""
import ctypes
import glob
import os
import subprocess

from improving_segmentation_with_selfsupervised_depth_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_SO = os.path.join(ROOT, "tests", "hipemu", "libsegsde_emu.so")


def _stale():
    if not os.path.exists(EMU_SO):
        return True
    t = os.path.getmtime(EMU_SO)
    srcs = glob.glob(os.path.join(ROOT, "improving_segmentation_with_selfsupervised_depth_amd", "csrc", "*")) + \
        glob.glob(os.path.join(ROOT, "include", "*.h")) + glob.glob(os.path.join(ROOT, "tests", "hipemu", "hip", "*.h"))
    return any(os.path.getmtime(s) > t for s in srcs)


def install():
    if _stale():
        subprocess.check_call([os.path.join(ROOT, "tests", "hipemu", "build_emu.sh"), EMU_SO])
    cdll = _lib.bind(ctypes.CDLL(EMU_SO))
    _lib._LIB = cdll
    _lib.HOST_POINTERS_OK = True
    return cdll
