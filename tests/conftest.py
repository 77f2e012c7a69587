import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The shared libraries are build artefacts (git-ignored): make sure they exist and are not older than
    their sources before any test loads them (hipcc cross-compiles gfx950 without a GPU)."""
    import __graft_entry__ as entry
    entry.build()


@pytest.fixture(scope="session")
def synth():
    return importlib.import_module("nid-pose-estimation_amd.synth")


@pytest.fixture(scope="session")
def capi():
    return importlib.import_module("nid-pose-estimation_amd.capi")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_py
    oracle_py.load()
    return oracle_py


@pytest.fixture(scope="session")
def pair_S(synth):
    return synth.make_pair("S")


@pytest.fixture(scope="session")
def pair_S_edge(synth):
    return synth.make_pair("S", edge_cases=True)


@pytest.fixture(scope="session")
def pair_A(synth):
    return synth.make_pair("A")


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """Which Jacobian comparisons passed ONLY on the reference's measured noise (tests/test_parity_gpu.py: NOISE_K *
    |J_oracle - J_twin|, capped), so that a green run says where the plain 1e-9 bound was not enough."""
    mod = sys.modules.get("test_parity_gpu") or sys.modules.get("tests.test_parity_gpu")
    passes = getattr(mod, "NOISE_PASSES", None) if mod else None
    if passes is None:
        return
    tr = terminalreporter
    tr.write_sep("-", f"Jacobian cells that passed on the reference-noise term: {len(passes)}")
    by_test = {}
    for tid, cell, dj, nz in passes:
        by_test.setdefault(tid, []).append((cell, dj, nz))
    for tid, rows in sorted(by_test.items()):
        worst = max(rows, key=lambda r: r[1])
        tr.write_line(f"  {tid}: {len(rows)} cell(s); worst |dJ| {worst[1]:.2e} of its own scale in cell {worst[0]} "
                      f"(measured noise {worst[2]:.2e} of it)")
    counts = getattr(mod, "CLAUSE_COUNTS", {})
    if counts:
        tot = {k: sum(d[k] for d in counts.values()) for k in ("plain", "condition", "noise", "masked")}
        tr.write_sep("-", "Jacobian cells by the clause they passed on: plain %(plain)d, condition %(condition)d, noise %(noise)d; masked (history-dependent) %(masked)d" % tot)
        by_fn = {}
        for tid, d in counts.items():
            fn = tid.split("[")[0]
            e = by_fn.setdefault(fn, {"plain": 0, "condition": 0, "noise": 0, "masked": 0})
            for k in e:
                e[k] += d[k]
        for fn, d in sorted(by_fn.items()):
            if d["condition"] or d["noise"] or d["masked"]:
                tr.write_line(f"  {fn}: plain {d['plain']}, condition {d['condition']}, noise {d['noise']}, masked {d['masked']}")
    hist = getattr(mod, "HISTORY_CELLS", [])
    if hist:
        tr.write_sep("-", f"cells whose REFERENCE Jacobian depends on its call history (class H): {len(hist)}")
        by_test = {}
        for tid, cell, dfz, dsf in hist:
            by_test.setdefault(tid, []).append((cell, dfz, dsf))
        for tid, rows in sorted(by_test.items()):
            w1, w2 = max(rows, key=lambda r: r[1]), max(rows, key=lambda r: r[2])
            tr.write_line(f"  {tid}: {len(rows)} cell(s); right-after-computeHref vs zero slots up to {w1[1]:.2e} of the cell's scale (cell {w1[0]}), "
                          f"after another evaluation vs right after computeHref up to {w2[2]:.2e} (cell {w2[0]}); the HIP path equals the zero-slot value within 1e-9")
    try:
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "noise_term_cells.txt"), "w") as f:
            f.write("# test id, cell, |dJ| / own scale, noise / own scale -- cells that needed the noise term\n")
            for row in passes:
                f.write("%s %d %.3e %.3e\n" % row)
        with open(os.path.join(out, "jacobian_clause_counts.txt"), "w") as f:
            f.write("# test id: cells that passed on the plain bound / on the condition clause / on the noise term; masked cells\n")
            for tid, d in sorted(counts.items()):
                f.write(f"{tid} plain {d['plain']} condition {d['condition']} noise {d['noise']} masked {d['masked']}\n")
            f.write("# class H (test_history_dependence_is_bounded): test id, cell, |J_first - J_zero| / scale, |J_second - J_first| / scale\n")
            for row in hist:
                f.write("%s %d %.3e %.3e\n" % row)
    except OSError:
        pass
