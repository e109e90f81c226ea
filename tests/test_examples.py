"""The batch examples and the console entry point (f2) under test.

examples/example_ukf_rts_smoother_batch.py and examples/example_gaussian_process_batch.py are run on the reference's
historical data file and compared with what the REFERENCE's two batch examples compute on it (tests/golden/
batch_examples.npz, made by tests/golden/make_golden.py ``example_cases``: the reference scripts' compute part run as
written, with the sphere distance / heading pair injected and the noise generator seeded per ship)."""
import importlib.util
import os
import stat
import subprocess
import sys

import numpy as np
import pytest
from conftest import GOLDEN, PKG_ROOT, ROOT

CSV = os.path.join(GOLDEN, "data", "historical_ship_data.csv.gz")


def _load_example(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "examples", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.gpu
def test_ukf_batch_example_vs_reference_example():
    """Ship selection (ids.pop(1), dt > 48 h skip, error -> continue) and per-ship results of the one-launch example
    against the reference's per-ship loop: the same 116 ids, the same fate for every one of them, and for every ship
    that came through, rows 0, N/2 and N of the filtered and smoothed histories with the same injected noise."""
    g = np.load(os.path.join(GOLDEN, "batch_examples.npz"))
    ex = _load_example("example_ukf_rts_smoother_batch")
    res = ex.run(CSV, out_path=None, sphere=True, seed_base=int(g["seed_base"]), verbose=False)
    assert res["ids"].tolist() == g["ids"].tolist() and len(res["ids"]) == 116
    assert "id.tidy" not in res["ids"].tolist()
    assert res["category"].tolist() == g["category"].tolist()
    assert (g["category"] == "skipped").sum() == 43 and (g["category"] == "error").sum() == 2
    pos = {int(p): k for k, p in enumerate(res["kept_pos"])}
    checked = 0
    for i, cat in enumerate(g["category"]):
        if cat != "ok":
            continue
        k = pos[i]
        rows = g[f"ukf_{i}_rows"]
        assert rows[-1] == res["nsteps"][k]
        for key in ("means", "means_smoothed"):
            ref = g[f"ukf_{i}_{key}"]
            err = np.max(np.abs(res[key][k, rows] - ref) / np.maximum(np.abs(ref), 1e-3))
            assert err < 1e-6, (res["ids"][i], key, err)
        for key in ("covs", "covs_smoothed"):
            ref = g[f"ukf_{i}_{key}"]
            err = np.max(np.abs(res[key][k, rows] - ref) / np.max(np.abs(ref), axis=(-1, -2), keepdims=True))
            assert err < 1e-5, (res["ids"][i], key, err)
        checked += 1
    assert checked == 71


@pytest.mark.gpu
def test_gp_batch_example_vs_reference_example():
    """Lock-step batched fits (2 seeded restarts, run as extra batch entries) of the first 8 ships against the reference
    example's per-ship scikit-learn fits with the same gpr_kwargs: fitted log marginal likelihood, theta and the
    predictions at the example's prediction times."""
    g = np.load(os.path.join(GOLDEN, "batch_examples.npz"))
    ex = _load_example("example_gaussian_process_batch")
    n = int(g["gp_count"])
    res = ex.run(CSV, out_path=None, restarts=2, random_state=0, max_ships=n, verbose=False)
    assert res["ids"].tolist() == g["ids"].tolist()[:n]
    for i in range(n):
        assert np.isclose(res["lml"][i], g[f"gp_{i}_lml"], rtol=1e-6), i
        np.testing.assert_allclose(res["thetas"][i], g[f"gp_{i}_theta"], rtol=2e-3, atol=2e-3)
        np.testing.assert_allclose(res[f"pred_{i}"], g[f"gp_{i}_pred"], rtol=1e-5, atol=1e-4)
        np.testing.assert_allclose(res[f"std_{i}"], g[f"gp_{i}_std"], rtol=1e-4, atol=1e-5)


def _entry_point():
    import tomli

    with open(os.path.join(ROOT, "pyproject.toml"), "rb") as f:
        meta = tomli.load(f)
    return meta, meta["project"]["scripts"]["track_estimator"]


def test_console_script_declared_like_the_reference():
    """pyproject.toml declares the reference's console script (reference pyproject.toml:41) and it resolves to a callable
    in this package."""
    meta, target = _entry_point()
    assert target == "track_estimators.cli.main_cli:track_estimator"
    assert meta["project"]["name"] == "track_estimators"
    mod, fn = target.split(":")
    m = importlib.import_module(mod)
    assert callable(getattr(m, fn)) and m.__file__.startswith(PKG_ROOT)


@pytest.mark.gpu
def test_cli_example_run_sh_command_line(tmp_path):
    """The command line of the reference's examples/cli_example/run.sh:2, unchanged, through a `track_estimator`
    executable generated from the pyproject entry point (what `pip install` would put on PATH), with the reference's
    input.json values: the six output files appear with the reference's names and shapes."""
    _, target = _entry_point()
    mod, fn = target.split(":")
    bindir = tmp_path / "bin"
    bindir.mkdir()
    exe = bindir / "track_estimator"
    exe.write_text(f"#!{sys.executable}\nimport sys\nsys.path.insert(0, {PKG_ROOT!r})\nfrom {mod} import {fn}\n"
                   f"sys.exit({fn}())\n")
    exe.chmod(exe.stat().st_mode | stat.S_IEXEC)
    work = tmp_path / "cli_example"
    data = tmp_path / "data" / "historical_ships"
    work.mkdir()
    data.mkdir(parents=True)
    import gzip
    import shutil

    with gzip.open(CSV, "rb") as src, open(data / "historical_ship_data.csv", "wb") as dst:
        shutil.copyfileobj(src, dst)
    (work / "input.json").write_text(
        '{"dim": 4, "H": [1, 1, 0, 0], "R": [0.001, 0.001, 0, 0], "Q": [1e-2, 1e-2, 1e-4, 1e-4], '
        '"P": [1.0, 1.0, 1.0, 1.0], "dt": -1, "nsteps": 2}')  # the values of the reference's input.json
    cmd = ('track_estimator -i input.json -o "output" -t ../data/historical_ships/historical_ship_data.csv -s 01203823 '
           '-ic "primary.id" -lat "lat" -lon "lon" -rts')
    env = dict(os.environ, PATH=f"{bindir}{os.pathsep}{os.environ['PATH']}")
    r = subprocess.run(["bash", "-c", cmd], cwd=work, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    pred = np.loadtxt(work / "output_01203823_predictions.txt")
    assert pred.shape == (103, 4)
    np.testing.assert_allclose(pred[0], [-30.5, -0.5, 14.578418614021368, 198.52495095065817], rtol=1e-11)
    for suffix in ("variances", "dts", "predictions_smoothed", "variances_smoothed"):
        assert (work / f"output_01203823_{suffix}.txt").exists(), suffix


@pytest.mark.gpu
def test_savgol_example_vs_reference_example():
    """examples/example_ukf_rts_smoother_savgol.py against the reference script of the same name run as written
    (tests/golden/savgol_example.npz, make_golden.savgol_case: ship 01205070 in reverse, Savitzky-Golay on SOG / COG,
    sphere pair injected, noise zeroed): the smoothed inputs exactly, the filtered and smoothed histories to the parity
    bound."""
    g = np.load(os.path.join(GOLDEN, "savgol_example.npz"))
    ex = _load_example("example_ukf_rts_smoother_savgol")
    res = ex.run(CSV, out_path=None, sphere=True, inject_noise=False)
    for key in ("sog", "cog", "z"):
        np.testing.assert_allclose(res[key], g[key], rtol=1e-13, atol=1e-13)
    assert np.array_equal(res["dt"], g["dt"]) and len(res["dt"]) == 66
    for key in ("means", "means_smoothed"):
        err = np.max(np.abs(res[key] - g[key]) / np.maximum(np.abs(g[key]), 1e-3))
        assert err < 1e-6, (key, err)
    for key in ("covs", "covs_smoothed"):
        err = np.max(np.abs(res[key] - g[key]) / np.max(np.abs(g[key]), axis=(-1, -2), keepdims=True))
        assert err < 1e-5, (key, err)


def test_savgol_example_host_preparation_vs_reference():
    """CPU half of the same: read_csv(reverse=True) + calculate_cog / calculate_sog + savgol_filter + measurements and
    rates of the host ShipTrack reproduce the reference's arrays exactly (no GPU needed up to the filter)."""
    from scipy.signal import savgol_filter
    from track_estimators.ship_track import ShipTrack
    from track_estimators.utils import haversine_formula, heading

    g = np.load(os.path.join(GOLDEN, "savgol_example.npz"))
    st = ShipTrack(calc_distance_func=haversine_formula, calc_heading_func=heading)
    st.read_csv(csv_file=CSV, ship_id="01205070", id_col="id", lat_col="lat", lon_col="lon", reverse=True)
    st.calculate_cog()
    st.calculate_sog()
    assert np.array_equal(st.sog, g["raw_sog"]) and np.array_equal(st.cog, g["raw_cog"])
    st.sog = savgol_filter(st.sog, 20, 4)
    st.cog = savgol_filter(st.cog, 4, 2)
    z = st.get_measurements(include_sog=True, include_cog=True)
    st.calculate_cog_rate()
    st.calculate_sog_rate()
    assert np.array_equal(z, g["z"]) and np.array_equal(st.dts, g["dts"])
    assert np.array_equal(st.sog_rate, g["sog_rate"]) and np.array_equal(st.cog_rate, g["cog_rate"])
