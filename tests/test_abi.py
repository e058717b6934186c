"""The C-ABI libraries load on a machine without a GPU and export every symbol include/pfac.h declares.
(No compute call here: without a GPU the HIP library must refuse, loudly, not fall back.)"""
import ctypes as C
import os
import re

import pytest

from phfpfac_amd import _ffi

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    hdr = open(os.path.join(REPO, "include", "pfac.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(pfac_[a-z0-9_]+)\s*\(", hdr)))


def test_every_declared_symbol_is_exported():
    host, hip = _ffi.host_lib(), _ffi.hip_lib()
    names = declared_symbols()
    assert len(names) >= 30
    missing = [n for n in names if not (hasattr(host, n) or hasattr(hip, n))]
    assert not missing, missing
    assert set(_ffi.HOST_SYMBOLS) | set(_ffi.HIP_SYMBOLS) == set(names)


def test_hip_code_object_is_gfx950_only():
    blob = open(os.path.join(_ffi.LIB_DIR, "libpfac_hip.so"), "rb").read()
    assert b"gfx950" in blob
    for other in (b"gfx942", b"gfx90a", b"gfx1100", b"sm_"):
        assert other not in blob


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from phfpfac_amd import GpuMatcher, PfacError
    with pytest.raises(PfacError) as e:
        GpuMatcher(0, 1)
    assert e.value.status in (_ffi.PFAC_E_NO_DEVICE, _ffi.PFAC_E_HIP)
    assert "no CPU fallback" in str(e.value) or "device" in str(e.value).lower()


def test_product_never_touches_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use oracle/."""
    pkg = os.path.join(REPO, "phfpfac_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".c", ".h", ".hip", "Makefile")):
                txt = open(os.path.join(root, f), errors="replace").read()
                assert "liboracle" not in txt and "pfac_oracle" not in txt and "import orc" not in txt, os.path.join(root, f)


def test_reference_seam_library_exports_the_reference_names():
    """libpfac_seam.so defines the three functions main.cc:35-37 declares, with C++ linkage and the reference's
    parameter lists (thread_data by value; the cudaStream_t is accepted as a void* and as a hipStream_t)."""
    lib = os.path.join(_ffi.LIB_DIR, "libpfac_seam.so")
    blob = open(lib, "rb").read()
    for mangled in (b"_Z17GPU_Malloc_Memory11thread_dataPPhPPiS3_PPjS3_S3_", b"_Z14GPU_TraceTable11thread_dataPvPhPiS2_PjS2_S2_",
                    # what main.cc:36 references once its cudaStream_t has become hipStream_t (= ihipStream_t *)
                    b"_Z14GPU_TraceTable11thread_dataP12ihipStream_tPhPiS3_PjS3_S3_",
                    b"_Z15GPU_Free_memoryPPhPPiS2_PPjS2_S2_"):
        assert mangled in blob
    hdr = open(os.path.join(REPO, "include", "pfac_seam.h")).read()
    for name in ("GPU_Malloc_Memory", "GPU_TraceTable", "GPU_Free_memory", "struct thread_data"):
        assert name in hdr
