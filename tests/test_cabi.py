"""The C-ABI library without a GPU: it loads, exports every symbol include/sd_frontend.h declares, and its
host-side entry points behave (no compute calls here)."""
import ctypes as C
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include/sd_frontend.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sd_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported(fe):
    L = fe.lib()
    names = declared_symbols()
    assert len(names) >= 40
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, "declared in sd_frontend.h but not exported: %s" % missing


def test_status_strings_and_version(fe):
    L = fe.lib()
    assert L.sd_version() >= 100
    for code in range(0, -7, -1):
        assert L.sd_status_string(code)


def test_invalid_arguments_are_reported_not_fatal(fe):
    L = fe.lib()
    h = C.c_void_p()
    assert L.sd_extractor_create(C.byref(h), 0, 1.2, 8, 20, 7) == fe.SD_ERR_INVALID        # nfeatures < 1
    assert L.sd_extractor_create(C.byref(h), 1000, 1.0, 8, 20, 7) == fe.SD_ERR_INVALID     # scale factor <= 1
    assert L.sd_extractor_create(C.byref(h), 1000, 1.2, 99, 20, 7) == fe.SD_ERR_INVALID    # too many levels
    assert L.sd_extractor_create(C.byref(h), 1000, 1.2, 8, 0, 7) == fe.SD_ERR_INVALID      # threshold 0
    assert L.sd_extractor_create(None, 1000, 1.2, 8, 20, 7) == fe.SD_ERR_INVALID
    assert L.sd_batch_create(C.byref(h), None, 640, 480, 1) == fe.SD_ERR_INVALID
    assert L.sd_batch_destroy(None) == fe.SD_OK and L.sd_extractor_destroy(None) == fe.SD_OK


def test_no_device_is_an_error_not_a_fallback(fe):
    """In the CPU container there is no GPU: batch creation must fail loudly (SD_ERR_NO_DEVICE)."""
    if fe.device_count() > 0:
        return
    ex = fe.ORBextractor(1000, 1.2, 8, 20, 7)
    try:
        fe.Batch(ex, 640, 480, 1)
    except fe.SdError as e:
        assert e.code == fe.SD_ERR_NO_DEVICE
    else:
        raise AssertionError("Batch() succeeded without a device")


def test_blur_taps_validation(fe):
    ex = fe.ORBextractor(1000, 1.2, 8, 20, 7)
    ex.set_blur_taps([18, 34, 49, 55, 49, 34, 18])       # the plain-rounding variant (sum 257) is allowed
    try:
        ex.set_blur_taps([60, 60, 60, 60, 60, 60, 60])
    except fe.SdError as e:
        assert e.code == fe.SD_ERR_INVALID
    else:
        raise AssertionError("taps summing to 420 accepted")


def test_host_mirror_header_compiles():
    """slam-dynamic_amd/host/ORBextractor.h (the class-API mirror) is valid C++ against the C ABI."""
    import shutil
    import subprocess
    import tempfile
    hdr = os.path.join(ROOT, "slam-dynamic_amd/host/ORBextractor.h")
    if not os.path.exists(hdr) or not shutil.which("g++"):
        return
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.cpp")
        open(src, "w").write('#include "%s"\nint main(){ return 0; }\n' % hdr)
        subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-I" + os.path.join(ROOT, "include"), src])
