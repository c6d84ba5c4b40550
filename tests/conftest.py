import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


# ---------------------------------------------------------------------------------------------- GEMM arithmetic of the step plans
# The whole-step plans run their encoder GEMMs on three exact bf16 terms per operand (six matrix instructions per k-step, no
# operand bit dropped: the library default since round 5) or, opt-in, on two f16 terms (functions.SplitGemm.f16 / RR_F16X2=1:
# 22 significant bits scaled per tensor, three instructions).  The modules that compare the HIP path with the oracle / the
# reference's vectors run in BOTH modes; the modules that compare two HIP paths bit for bit (the per-op Python mirror's
# bounds come from separate passes) are pinned to the three-term form.
BOTH_GEMM_MODES = {"test_gpu_model", "test_gpu_headline_kernels", "test_gpu_trainers", "test_gpu_api_parity"}
DEFAULT_MODE_ONLY = {"test_train_mode_plan_path_above_8192_rows_against_fp64_oracle_h600_d6",
                     "test_train_mode_plan_path_above_8192_rows_against_fp64_oracle_h300"}
BF16X3_ONLY = {"test_gpu_plan", "test_gpu_dp_trainers", "test_gpu_split"}      # (test_gpu_cxx_host sets the arithmetic itself)


def pytest_generate_tests(metafunc):
    mod = metafunc.module.__name__.rsplit(".", 1)[-1]
    if mod in BOTH_GEMM_MODES and "gemm_mode" in metafunc.fixturenames:
        # (the two train-mode steps above 8,192 rows against the fp64 oracle take 1-2 minutes of CPU time each: the default
        # arithmetic only - the two-term form at those sizes is held by the GEMM-level tests of the same module, by
        # test_full_step_size_properties and by tests/test_gpu_f16x2.py's plan-against-plan comparison; its measured numbers
        # from round 4's runs stay in profiles/r04_parity_errors.txt)
        modes = ["bf16x3"] if metafunc.function.__name__ in DEFAULT_MODE_ONLY else ["bf16x3", "f16x2"]
        metafunc.parametrize("gemm_mode", modes, indirect=True)


@pytest.fixture(autouse=True)
def gemm_mode(request):
    mod = request.module.__name__.rsplit(".", 1)[-1]
    mode = getattr(request, "param", "bf16x3" if mod in BF16X3_ONLY else None)
    if mode is None or request.node.get_closest_marker("gpu") is None:
        yield mode
        return
    from reactranker_amd import functions as Fn
    old = Fn.SplitGemm.f16
    Fn.SplitGemm.f16 = mode == "f16x2"
    try:
        yield mode
    finally:
        Fn.SplitGemm.f16 = old


# ---------------------------------------------------------------------------------------------- measured parity errors
# Every `-m gpu` parity test reports what it MEASURED (not just that it stayed under its bound) through the `parity_log`
# fixture; the lines are merged into gpurun_out/parity_errors.txt (one block per test id, the latest run wins) and the
# builder copies that file to profiles/rNN_parity_errors.txt.  $RR_PARITY_LOG overrides the path.
_PARITY = {}


def _parity_path():
    return os.environ.get("RR_PARITY_LOG", os.path.join(REPO, "gpurun_out", "parity_errors.txt"))


@pytest.fixture
def parity_log(request):
    lines = _PARITY.setdefault(request.node.nodeid, [])
    del lines[:]

    def log(msg):
        lines.append(str(msg))
        print("[parity] " + str(msg))
    return log


@pytest.fixture(autouse=True)
def _parity_table(request):
    """Collects what the close() / _compare() helpers measured during a -m gpu test (tests/helpers.py: record)."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    from tests import helpers
    table = {}
    helpers.CURRENT = table
    try:
        yield
    finally:
        helpers.CURRENT = None
        if table:
            lines = _PARITY.setdefault(request.node.nodeid, [])
            # labels often carry a case name: fold them into at most 12 lines, worst first
            items = sorted(table.items(), key=lambda kv: -kv[1][0])
            for what, (err, tol) in items[:12]:
                lines.append(f"max err {err:.3e}" + (f" (bound {tol:g})" if tol is not None else "") + f"  [{what}]")
            if len(items) > 12:
                lines.append(f"... and {len(items) - 12} more labels, all below {items[12][1][0]:.3e}")


def pytest_sessionfinish(session, exitstatus):
    if not _PARITY:
        return
    path = _parity_path()
    merged = {}
    if os.path.exists(path):
        cur = None
        with open(path) as f:
            for ln in f:
                ln = ln.rstrip("\n")
                if ln.startswith("## "):
                    cur = ln[3:]
                    merged[cur] = []
                elif cur is not None and ln.startswith("   "):
                    merged[cur].append(ln[3:])
    merged.update({k: v for k, v in _PARITY.items() if v})
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        f.write("# measured errors of the -m gpu parity tests (tests/conftest.py: parity_log); latest run per test id\n")
        for k in sorted(merged):
            f.write("## " + k + "\n")
            for ln in merged[k]:
                f.write("   " + ln + "\n")
