import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

EMU_DIR = os.path.join(ROOT, "tests", "emu")
EMU_LIB = os.path.join(EMU_DIR, "libsvb_emu.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


def _build_emu():
    """Compile the unmodified kernel sources against the CPU lane emulator (test infrastructure)."""
    r = subprocess.run([os.path.join(EMU_DIR, "build_emu.sh")], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("emulator build failed:\n" + r.stdout + r.stderr)


@pytest.fixture(scope="session")
def _emu_lib():
    _build_emu()
    from neuralsvb_amd import _lib
    return _lib.bind(EMU_LIB)


@pytest.fixture(params=["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
def dev(request):
    """Device under test.

    'gpu': the product path -- libsvb_hip.so on cuda:0 (fails loudly if the extension or GPU is missing).
    'emu': CPU tensors + the lane-level emulator build of the same kernel sources; exists because the build
           container has no GPU.  It is injected here, by the test harness only.
    """
    from neuralsvb_amd import _lib
    if request.param == "gpu":
        assert torch.cuda.is_available(), "gpu-marked test needs an MI355X"
        _lib._LIB, _lib._LIB_IS_EMU = None, False
        _lib.get_lib()  # raises SvbLibraryMissing if not built
        yield torch.device("cuda:0")
    else:
        lib = request.getfixturevalue("_emu_lib")
        old = (_lib._LIB, _lib._LIB_IS_EMU)
        _lib._LIB, _lib._LIB_IS_EMU = lib, True
        try:
            yield torch.device("cpu")
        finally:
            _lib._LIB, _lib._LIB_IS_EMU = old


@pytest.fixture
def gpu_only():
    assert torch.cuda.is_available(), "gpu-marked test needs an MI355X"
    from neuralsvb_amd import _lib
    _lib._LIB, _lib._LIB_IS_EMU = None, False
    _lib.get_lib()
    return torch.device("cuda:0")
