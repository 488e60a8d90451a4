"""CPU-only checks of the drop-in boundary: the shared library loads without a GPU, exports every entry point
include/po_hip.h declares, the ctypes struct mirror has the C layout, host-side helpers agree with the oracle,
and compute calls fail loudly (no CPU fallback) when no device is present."""
import ctypes
import os
import re
import subprocess
import tempfile

import numpy as np
import pytest

from path_optimizer_amd import abi, binding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "po_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(po_[a-z_0-9]+)\s*\(", hdr))
    assert {"po_create", "po_solve_batch", "po_solve_batch_device", "po_destroy", "po_assemble_batch", "po_scaling_batch"} <= names
    L = binding.lib()
    for n in sorted(names):
        assert hasattr(L, n), f"{n} declared in include/po_hip.h but not exported"
    assert set(binding.EXPORTS) <= names


def test_struct_layouts_match_c():
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "po_hip.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(po_params), sizeof(po_info), sizeof(po_batch_in), sizeof(po_batch_out), offsetof(po_params, eps_abs), offsetof(po_params, max_iter));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")])
        out = subprocess.check_output([os.path.join(d, "t")]).split()
    got = [int(x) for x in out]
    assert got[:4] == [ctypes.sizeof(abi.PoParams), ctypes.sizeof(abi.PoInfo), ctypes.sizeof(abi.PoBatchIn), ctypes.sizeof(abi.PoBatchOut)]
    assert got[4] == abi.PoParams.eps_abs.offset and got[5] == abi.PoParams.max_iter.offset
    assert np.dtype(abi.INFO_DTYPE).itemsize == ctypes.sizeof(abi.PoInfo)


def test_host_helpers_agree_with_oracle(oracle):
    po, pd = oracle.default_params(), binding.default_params()
    for name, _ in abi.PoParams._fields_:
        a, b = getattr(po, name), getattr(pd, name)
        assert (list(a) == list(b)) if name == "d" else (a == b), name
    for form in (0, 1, 2):
        for N, keep in ((2, 4), (80, 4), (200, 4), (77, 3 if form == 0 else 4)):
            k = 4 if form == 1 else keep
            assert binding.problem_dims(form, N, k) == oracle.dims(form, N, k)
    for ds in (0.15, 0.25, 0.3, 0.5, 1.0, 2.0):
        s = ds * np.arange(40)
        assert binding.keep_control_steps(0, s) == oracle.keep_steps(0, s)
    with pytest.raises(binding.PoError):
        binding.problem_dims(0, 1, 4)
    with pytest.raises(binding.PoError):
        binding.problem_dims(1, 50, 3)  # KPC hard-codes keep = 4


def test_no_cpu_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(binding.PoError):
        binding.Engine(0)  # po_create fails loudly: there is no CPU path behind the C ABI


def test_host_mirror_compiles():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "path_optimizer_amd", "host")], stdout=subprocess.DEVNULL)
    assert os.path.exists(os.path.join(ROOT, "path_optimizer_amd", "host", "host_test"))


def test_inline_dpp_fmacs_of_the_tension_solve_have_their_wait_states():
    """po_smooth.hip's blocked substitution broadcasts the block vector inside v_fmac_f64_dpp (row_newbcast) — inline assembly, invisible to the compiler's hazard recogniser.
    Compile the file to ISA and check that no VALU write of a DPP source register sits within two instructions in front of such an FMAC (tools/dpp_hazard_check.py)."""
    import shutil
    import sys

    if not shutil.which("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc here")
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from dpp_hazard_check import check

    total, findings = check()
    assert total >= 100 and not findings, (total, findings[:3])
    assert check.led * 9 == total, (check.led, total)  # one wait-state-carrying FMAC per product of nine (bandwidth W = 9)


def test_library_version_is_the_headers_abi_and_the_binding_checks_it():
    """po_version() is built from PO_ABI_VERSION (not a literal), the header, abi.py and the loaded library agree, and binding.lib() refuses a library of another ABI
    (ABI 5 removed fields from the middle of po_params; a stale libpo_hip.so would have been driven with a shifted struct)."""
    hdr = open(os.path.join(ROOT, "include", "po_hip.h")).read()
    m = re.search(r"#define\s+PO_ABI_VERSION\s+(\d+)", hdr)
    assert m and int(m.group(1)) == abi.PO_ABI_VERSION
    ver = binding.lib().po_version().decode()
    assert ver == f"po_hip {abi.PO_ABI_VERSION} (gfx950)", ver
    assert re.search(r"#define\s+PO_NOT_AVAILABLE\s+\(-2\)", hdr) and abi.PO_NOT_AVAILABLE == -2
    # a binding written against another ABI refuses this library at load time
    code = ("import sys; sys.path.insert(0, %r); from path_optimizer_amd import abi; abi.PO_ABI_VERSION = 5; from path_optimizer_amd import binding; binding.lib()" % ROOT)
    r = subprocess.run([os.sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "PO_ABI_VERSION 5" in r.stderr, r.stderr[-600:]


def test_po_create_refuses_round_counts_that_would_switch_the_refinement_off():
    """refine_rounds + refine_extra_rounds >= 32 does not fit the hand-back status: it used to run the plain solve silently; now PO_ERR_INVALID — before any
    device is touched, so the check runs on the CPU box too."""
    p = binding.default_params()
    p.refine = 2; p.refine_rounds = 30; p.refine_extra_rounds = 2
    h = ctypes.c_void_p()
    assert binding.lib().po_create(0, ctypes.byref(p), ctypes.byref(h)) == abi.PO_ERR_INVALID
    p.refine_rounds = 5  # (the headline setting: 5 + 2 rounds) passes the parameter check — what follows is the device check
    rc = binding.lib().po_create(0, ctypes.byref(p), ctypes.byref(h))
    assert rc in (abi.PO_OK, abi.PO_ERR_HIP)
    if rc == abi.PO_OK:
        binding.lib().po_destroy(h)
