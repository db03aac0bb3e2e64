import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

EMU_LIB = os.path.join(ROOT, "tests", "hipemu", "libpfv_emu.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def build_emulator() -> str:
    """g++ build of the UNMODIFIED csrc/ sources against tests/hipemu (CPU fiber emulator)."""
    csrc = os.path.join(ROOT, "pretty-fast-video_amd", "csrc")
    emu = os.path.join(ROOT, "tests", "hipemu")
    srcs = [os.path.join(csrc, f) for f in os.listdir(csrc) if os.path.isfile(os.path.join(csrc, f))] + [os.path.join(emu, "hipemu.cpp"),
                                                                os.path.join(emu, "hip", "hip_runtime.h"),
                                                                *[os.path.join(ROOT, "include", h) for h in ("pfv_hip.h", "pfv_hip_core.h", "pfv_hip_ext.h")]]
    defs = os.environ.get("PFV_EMU_DEFS", "").split()      # developer switch: e.g. -DPFV_PENC_PERSISTENT
    lib = EMU_LIB if not defs else EMU_LIB.replace(".so", "_" + "".join(c for c in "".join(defs) if c.isalnum()) + ".so")
    if os.path.exists(lib) and all(os.path.getmtime(s) <= os.path.getmtime(lib) for s in srcs):
        return lib
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", *defs, "-I", emu, "-x", "c++",
                    os.path.join(csrc, "pfv_capi.hip"), os.path.join(emu, "hipemu.cpp"), "-o", lib], check=True)
    return lib


@pytest.fixture(scope="session")
def graft():
    import __graft_entry__ as g
    return g


@pytest.fixture(scope="session")
def pkg(graft):
    return graft.load_package()


@pytest.fixture(scope="session")
def oracle():
    from oracle_bind import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def gpu_ctx(graft, pkg):
    """context on the real GPU through the in-tree libpfv_hip.so (no emulator, no fallback)"""
    import libswitch
    if os.environ.get("PFV_TEST_EMU_AS_GPU") == "1":
        # developer dry-run of the -m gpu tests in the GPU-less build container (never set by the driver)
        libswitch.use(pkg, build_emulator())
    else:
        libswitch.reset(pkg)
        graft.build_hip()
    ctx = pkg.Context(0)
    yield ctx
    ctx.close()


@pytest.fixture(scope="session")
def emu_ctx(graft, pkg):
    """context on the CPU emulator build of the same kernel sources (logic check only)"""
    import libswitch
    libswitch.use(pkg, build_emulator())
    ctx = pkg.Context(0)
    yield ctx
    ctx.close()
    libswitch.reset(pkg)
