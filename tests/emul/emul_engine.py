"""TEST-ONLY helper: an Engine bound to the CPU emulation build of the kernel sources (hip_emul.h)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
EMUL_LIB = os.path.join(_HERE, "libscanfold_emul.so")


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def emul_engine(paramset=None):
    from scanfold_amd._lib import Engine
    build()
    return Engine(device=0, paramset=paramset, lib_path=EMUL_LIB)
