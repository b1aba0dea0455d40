import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The shared library is a build product (git-ignored): from a clean checkout build it once with hipcc (cross-compiles
    gfx950 without a GPU, ~35 s) so the ABI tests have something to load.  A missing hipcc leaves the tests to fail loudly."""
    import shutil
    import subprocess
    lib = os.path.join(ROOT, "instantir_amd", "libinstantir_hip.so")
    if not os.path.exists(lib) and shutil.which("hipcc"):
        subprocess.run([os.path.join(ROOT, "instantir_amd", "csrc", "build.sh")], check=False, stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def record_psnr(name, value):
    """Append a measured PSNR to gpurun_out/psnr.log (merged back from the GPU box) so stated bounds can cite measurements."""
    d = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "psnr.log"), "a") as f:
            f.write(f"{name} {value:.2f}\n")
    except OSError:
        pass
