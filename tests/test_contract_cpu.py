"""Static checks of two rules of the build brief that no device test can see.

1. The oracle is test infrastructure: nothing in the product packages (`vcr_gaus_amd/`, `diff_gaussian_rasterization/`), in
   `examples/` or in `bench.py` outside its `cpu_*` baseline functions imports `oracle`; nothing anywhere in the product reads
   `/root/reference`; the binding has no fallback when the HIP library is missing.
2. The bench line: the committed records of the driver's command carry every key of the contract, with consistent arithmetic
   (value = ranks x steps / time, roofline.frac = achieved / peak = algorithmic bytes / launch time / peak)."""
import ast
import glob
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _imports(tree):
    """(module name, enclosing top-level function or None) of every import in the tree"""
    out = []

    def walk(node, fn):
        for child in ast.iter_child_nodes(node):
            inner = child.name if isinstance(child, (ast.FunctionDef, ast.AsyncFunctionDef)) and fn is None else fn
            if isinstance(child, ast.Import):
                out.extend((a.name, fn) for a in child.names)
            elif isinstance(child, ast.ImportFrom):
                out.append((child.module or "", fn))
            walk(child, inner)
    walk(tree, None)
    return out


def _sources(*dirs):
    files = []
    for d in dirs:
        files += glob.glob(os.path.join(ROOT, d, "**", "*.py"), recursive=True)
    return sorted(files)


def test_product_never_imports_the_oracle():
    for path in _sources("vcr_gaus_amd", "diff_gaussian_rasterization", "examples"):
        for mod, _ in _imports(ast.parse(open(path).read())):
            assert mod.split(".")[0] != "oracle", f"{path} imports {mod}"
            assert not mod.startswith("tests"), f"{path} imports {mod}"


def test_bench_touches_the_oracle_only_in_its_cpu_baseline():
    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    users = {fn for mod, fn in _imports(tree) if mod.split(".")[0] == "oracle"}
    assert users, "bench.py is expected to time the oracle as its cpu_baseline"
    assert None not in users, "bench.py imports the oracle at module level"
    assert all(fn.startswith("cpu_") for fn in users), users


def test_nothing_that_travels_reads_the_reference_tree():
    # (tests/golden/make_golden*.py run in the build container only: they are the committed generators of the fixtures)
    files = _sources("vcr_gaus_amd", "diff_gaussian_rasterization", "examples", "oracle") + \
        [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")] + \
        [p for p in _sources("tests") if os.sep + "golden" + os.sep not in p and os.path.basename(p) != "test_contract_cpu.py"]
    for path in files:
        src = open(path).read()
        tree = ast.parse(src)
        for node in ast.walk(tree):
            if isinstance(node, ast.Constant) and isinstance(node.value, str) and "/root/reference" in node.value:
                # docstrings may CITE the path; a string that is used as data may not
                doc_owner = [n for n in ast.walk(tree) if isinstance(n, (ast.Module, ast.FunctionDef, ast.ClassDef))
                             and ast.get_docstring(n, clean=False) == node.value]
                assert doc_owner, f"{path} holds the reference path in a non-docstring literal"


def test_binding_has_no_fallback():
    src = open(os.path.join(ROOT, "vcr_gaus_amd", "_lib.py")).read()
    tree = ast.parse(src)
    for node in ast.walk(tree):
        if isinstance(node, ast.Try):          # a handler around the load may re-raise, never continue on something else
            for h in node.handlers:
                assert any(isinstance(n, ast.Raise) for n in ast.walk(h)), "an except clause in _lib.py swallows the error"


def _records():
    names = ["r5_bench_final_tree.json", "r5_bench_driver_form.json"]
    return [os.path.join(ROOT, "profiles", n) for n in names if os.path.exists(os.path.join(ROOT, "profiles", n))]


@pytest.mark.parametrize("path", _records(), ids=os.path.basename)
def test_committed_bench_line_keeps_the_contract(path):
    line = json.loads(open(path).read().strip().splitlines()[-1])
    for key, typ in [("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                     ("config", dict), ("roofline", dict), ("cpu_baseline", dict)]:
        assert isinstance(line[key], typ), key
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert line["n_gpus"] == 1 and line["scaling"] == "weak" and line["higher_is_better"] is True
    assert line["vs_baseline"] is None              # BASELINE.md holds no published number for this metric
    assert line["dtype"] == "f32" and line["data"] == "synthetic"
    assert "workload" in line["config"] and "model" not in line["config"]
    assert line["config"]["gaussians"] == 1_000_000 and (line["config"]["width"], line["config"]["height"]) == (1920, 1080)
    assert "iters/s" in line["unit"] and "iters/sec" in json.dumps(base)
    # value = ranks x steps / wall time of the K steps
    assert line["value"] == pytest.approx(line["n_gpus"] * 1e3 / line["ms_per_step"], rel=1e-9)
    roof = line["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == pytest.approx(8000.0)
    assert roof["frac"] == pytest.approx(roof["achieved"] / roof["peak"], rel=1e-12)
    assert roof["achieved"] == pytest.approx(roof["algorithmic_bytes"] / (roof["avg_ms"] * 1e-3) / 1e9, rel=1e-9)
    # B6 = 60 R + 52 P for the metric configuration
    R, P = line["config"]["tile_instances_R"], 1920 * 1080
    assert roof["algorithmic_bytes"] == 60 * R + 52 * P
    assert roof["algorithmic_bytes_emitted"] == 60 * line["config"]["emitted_instances"] + 52 * P
    assert 0.0 < roof["frac_emitted"] <= roof["frac"] < 1.0
    assert roof["traffic"] is None or roof["traffic"] > 0
    bwd = line["roofline_bwd"]
    assert bwd["frac"] == pytest.approx(bwd["algorithmic_bytes"] / (bwd["avg_ms"] * 1e-3) / 1e9 / bwd["peak"], rel=1e-9)
    assert bwd["algorithmic_bytes"] == 60 * R + 84 * P + 120 * line["config"]["visible_V"]
    cpu = line["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cpu, key
    assert cpu["kind"] == "port" and cpu["cores"] >= 1 and cpu["value"] > 0
    assert line["value_densify_amortised"] == pytest.approx(line["schedule_inclusive"]["iters_per_s"])
    assert line["value_densify_amortised"] < line["value"]
