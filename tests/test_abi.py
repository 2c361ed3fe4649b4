"""The C-ABI library loads without a GPU and exports exactly what include/t2fit.h declares; the
ctypes mirror of its structs has the C layout.  No compute entry point is called here."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(REPO, "include", "t2fit.h")


@pytest.fixture(scope="module")
def lib():
    from fetal_t2mapping_amd import build
    from fetal_t2mapping_amd._lib import load

    build.build()  # hipcc cross-compiles gfx950 without a GPU
    return load()


def _declared():
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    return sorted(set(re.findall(r"\b(t2fit_[a-z_0-9]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from fetal_t2mapping_amd import _abi

    declared = _declared()
    assert len(declared) >= 11
    assert sorted(n for n, _, _ in _abi.SYMBOLS) == declared
    for name in declared:
        assert getattr(lib, name) is not None
    # nothing else with the prefix leaks out of the shared object
    out = subprocess.run(["nm", "-D", "--defined-only", lib._name], capture_output=True, text=True).stdout
    exported = sorted(set(re.findall(r"\b(t2fit_[a-z_0-9]+)$", out, flags=re.M)))
    assert exported == declared


def test_struct_layout_matches_the_header(tmp_path):
    from fetal_t2mapping_amd import _abi

    prog = tmp_path / "sz.c"
    prog.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "t2fit.h"\nint main(void){printf("%zu %zu %zu %zu %zu %d\\n",'
                    'sizeof(t2fit_config), offsetof(t2fit_config, te_ms), offsetof(t2fit_config, x0), '
                    'offsetof(t2fit_config, noprior_t2_ub), sizeof(t2fit_maps), T2FIT_ABI_VERSION);return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(REPO, "include"), "-o", str(exe), str(prog)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    cfg = _abi.T2FitConfig
    assert got == [C.sizeof(cfg), cfg.te_ms.offset, cfg.x0.offset, cfg.noprior_t2_ub.offset,
                   C.sizeof(_abi.T2FitMaps), _abi.ABI_VERSION]


def test_config_tables_and_argument_errors_without_gpu(lib):
    from fetal_t2mapping_amd import _abi

    cfg = _abi.T2FitConfig()
    assert lib.t2fit_abi_version() == _abi.ABI_VERSION
    assert lib.t2fit_config_default(C.byref(cfg), 7, 1) == _abi.E_INVALID
    assert b"model" in lib.t2fit_last_error()
    assert lib.t2fit_config_default(C.byref(cfg), _abi.MODEL_GAUSSIAN_RICIAN, 0) == _abi.OK
    assert (cfg.x0[0], cfg.lb[1], cfg.ub[0], cfg.ftol, cfg.maxls) == (890.0, 30.0, 30000.0, 1e-2, 50)
    assert lib.t2fit_device_count() >= 0
    # argument validation happens before any device work
    maps = _abi.T2FitMaps()
    assert lib.t2fit_volume_dev(None, None, 0, None, 10, C.byref(maps), None) == _abi.E_INVALID
    cfg.n_te = 1
    assert lib.t2fit_volume_dev(C.byref(cfg), C.c_void_p(8), 0, None, 10, C.byref(maps), None) == _abi.E_INVALID
    assert b"n_te" in lib.t2fit_last_error()


def test_product_has_no_cpu_path():
    """Without a HIP device every compute entry point of the host mirror raises; the product never imports the
    test oracle (checked in a fresh interpreter: this test process has the oracle loaded by other tests)."""
    import subprocess

    import numpy as np

    import fetal_t2mapping_amd as t2
    from fetal_t2mapping_amd._lib import load

    if load().t2fit_device_count() == 0:
        with pytest.raises(RuntimeError, match="no HIP device"):
            t2.fit_volume(np.ones((3, 1, 1, 4), np.float32), None, [114.0, 202.0, 299.0], "gaussian",
                          t2.fit_table("gaussian", True))
    pkg = os.path.join(REPO, "fetal_t2mapping_amd")
    for name in os.listdir(pkg):
        if name.endswith(".py"):
            src = open(os.path.join(pkg, name)).read()
            assert "import oracle" not in src and "from oracle" not in src, name
    code = ("import sys; sys.path.insert(0, %r); import fetal_t2mapping_amd, fetal_t2mapping_amd.cli, "
            "fetal_t2mapping_amd.dist, fetal_t2mapping_amd.stream; "
            "assert not [m for m in sys.modules if m == 'oracle' or m.startswith('oracle.')], 'oracle imported'" % REPO)
    subprocess.run([sys.executable, "-c", code], check=True)


def test_missing_library_fails_loudly():
    """No library, no fit: with T2FIT_LIB pointing nowhere the loader raises instead of falling back to anything
    (runs on the GPU box as well as here: the property matters where the product runs)."""
    import subprocess

    code = ("import sys; sys.path.insert(0, %r)\n"
            "import numpy as np, fetal_t2mapping_amd as t2\n"
            "try:\n"
            "    t2.fit_volume(np.ones((3, 1, 1, 4), np.float32), None, [114.0, 202.0, 299.0], 'gaussian',\n"
            "                  {'initial_guess': [650, 165], 'param_bounds': [(600, 10000), (10, 600)], 'solver': 'L-BFGS-B', 'options': {}})\n"
            "except RuntimeError as e:\n"
            "    assert 'is missing' in str(e) and 'no CPU fallback' in str(e), e\n"
            "    print('raised')\n" % REPO)
    env = dict(os.environ, T2FIT_LIB="/nonexistent/libt2fit_hip.so")
    out = subprocess.run([sys.executable, "-c", code], check=True, capture_output=True, text=True, env=env)
    assert out.stdout.strip() == "raised"
