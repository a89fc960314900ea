import os
import sys

import pytest

# torch bundles its own HIP runtime (libamdhip64).  It has to be the FIRST HIP runtime loaded
# into the process: if libmi_rt.so pulls in /opt/rocm's copy first, torch.cuda later reports
# "No HIP GPUs are available".  bench.py imports torch first for the same reason.
import torch  # noqa: F401,E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_library():
    """The HIP library is a build product (git-ignored).  From a clean checkout, build it once with the same
    script __graft_entry__.build() uses (hipcc cross-compiles gfx950 without a GPU); a box without hipcc
    keeps whatever travelled with the snapshot and fails loudly in abi.load() if there is nothing."""
    import shutil
    import subprocess
    lib = os.path.join(ROOT, "cs397raytracingsp22_amd", "lib", "libmi_rt.so")
    if not os.path.exists(lib) and (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        subprocess.run(["bash", os.path.join(ROOT, "cs397raytracingsp22_amd", "csrc", "build.sh")], check=True)
    yield


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (oracle/_build/liborc.so), built on demand with gcc."""
    from oracle import orc_py
    orc_py.build()
    orc_py.load()
    return orc_py


@pytest.fixture(scope="session")
def gpu_ctx():
    """One mi_ctx on GPU 0 through the C ABI.  Fails (does not skip) when the HIP library
    or the device is missing: -m gpu tests must exercise the native path."""
    from cs397raytracingsp22_amd import Context
    ctx = Context(0)
    yield ctx
    ctx.close()
