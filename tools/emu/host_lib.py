"""Loads translation units of the HIP library built for the host (build_host_lib.py) behind the ctypes
declarations of openvslam_amd._lib, so that tests written against the C ABI can run the library's own host
code and kernels without a GPU.  Test infrastructure: nothing in the package imports this module.

As a pytest plugin (`-p host_lib` with tools/emu on PYTHONPATH, or tools/emu/run_emulated.py) it points
openvslam_amd._lib at the emulated library for the whole session.
"""
import ctypes as C
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import build_host_lib  # noqa: E402


class _Missing:
    """A symbol of the C ABI that the emulated library does not hold."""

    def __init__(self, name):
        self.__dict__["name"] = name

    def __setattr__(self, k, v):
        self.__dict__[k] = v

    def __call__(self, *a):
        raise RuntimeError("%s is not part of the emulated library (translation unit not built for the host)" % self.name)


class EmulatedLibrary:
    def __init__(self, cdll):
        self.__dict__["_cdll"] = cdll
        self.__dict__["_missing"] = {}

    def __getattr__(self, name):
        try:
            return getattr(self._cdll, name)
        except AttributeError:
            return self._missing.setdefault(name, _Missing(name))


def load(units=None, work=None, out=None):
    from openvslam_amd import _lib
    units = units or build_host_lib.DEFAULT_UNITS
    work = work or os.environ.get("OVS_EMU_WORK", "/tmp/ovs_emu")
    out = out or os.path.join(work, "libovs_host_emu.so")
    build_host_lib.build(out, work, units)
    lib = EmulatedLibrary(C.CDLL(out))
    _lib._declare(lib)
    return lib


def install(units=None):
    """openvslam_amd._lib.lib() returns the emulated library from now on (this process only)."""
    from openvslam_amd import _lib
    lib = load(units)
    _lib._lib = lib
    return lib


def torch_on_host():
    """Tests keep arrays for the device-pointer entry points in torch tensors on "cuda".  Under emulation device
    memory is host memory, so a "cuda" device becomes the CPU and the handful of torch.cuda calls the tests make
    (streams, synchronisation) become no-ops: the pointers the library gets are then plain host pointers."""
    import contextlib

    import torch

    real_device = torch.device

    class FakeStream:
        _next = 0x1000

        def __init__(self, *a, **k):
            FakeStream._next += 0x10
            self.cuda_stream = FakeStream._next

        def wait_stream(self, other):
            pass

        def synchronize(self):
            pass

    class NullStream(FakeStream):
        def __init__(self):
            self.cuda_stream = 0

    def device(*a, **k):
        if a and (a[0] == "cuda" or (isinstance(a[0], str) and a[0].startswith("cuda"))):
            return real_device("cpu")
        return real_device(*a, **k)

    torch.device = device
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.current_stream = lambda *a, **k: NullStream()
    torch.cuda.Stream = FakeStream
    torch.cuda.stream = lambda s: contextlib.nullcontext()
    torch.cuda.mem_get_info = lambda *a, **k: (256 << 30, 256 << 30)


def pytest_configure(config):
    units = os.environ.get("OVS_EMU_UNITS")
    install(units.split(",") if units else None)
    torch_on_host()
