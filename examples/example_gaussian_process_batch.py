#!/usr/bin/env python3
"""
Batch Gaussian-process regression of every ship of a data file, on one MI355X.

Counterpart of the reference's examples/example_gaussian_process_batch.py:15-55: same ship selection, same kernel
``1.0 * RBF() + WhiteKernel(noise_level=0.5)`` (:41), same prediction times (:47-50); the per-ship loop of fits is
replaced by lock-step batched fits (``GPRegression.fit_batch``).  Headless: writes an ``.npz`` instead of the PDF.

    python examples/example_gaussian_process_batch.py [historical_ship_data.csv[.gz]] [out.npz] [n_restarts]
"""
import os
import sys
import time

import numpy as np
import pandas as pd
from sklearn.gaussian_process.kernels import RBF, WhiteKernel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ship-track-estimators_amd"))

from track_estimators.gaussian_processes.gaussian_process import GPRegression  # noqa: E402
from track_estimators.ship_track import ShipTrack  # noqa: E402
from track_estimators.utils import generate_dts  # noqa: E402


def run(csv, out_path=None, restarts=50, random_state=None, max_ships=None, verbose=True):
    """The compute part of the example; ``restarts`` / ``random_state`` go to the regressor exactly like the reference's
    ``gpr_kwargs`` (its default is 50 unseeded restarts, gaussian_process.py:50)."""
    df = pd.read_csv(csv)
    ids = df["primary.id"].unique().tolist()
    ids.pop(1)
    if max_ships is not None:
        ids = ids[:max_ships]
    tracks, kept = [], []
    for sid in ids:
        try:
            st = ShipTrack()
            st.read_csv(csv_file=csv, ship_id=sid, id_col="primary.id", lat_col="lat", lon_col="lon2", reverse=False)
        except Exception as exc:
            if verbose:
                print("Error in ", sid, type(exc).__name__)
            continue
        tracks.append(st)
        kept.append(sid)
    gp = GPRegression(kernel=1.0 * RBF() + WhiteKernel(noise_level=0.5))
    t0 = time.perf_counter()
    kw = {"n_restarts_optimizer": restarts}
    if random_state is not None:
        kw["random_state"] = random_state
    thetas, lml = gp.fit_batch(tracks, gpr_kwargs=kw)
    times = [np.insert(np.cumsum(generate_dts(st.dts, substeps=1)), 0, 0) for st in tracks]
    preds = gp.predict_batch(times)
    t1 = time.perf_counter()
    res = dict(ids=np.array([str(s) for s in kept]), thetas=thetas, lml=lml,
               **{f"pred_{i}": p[0] for i, p in enumerate(preds)}, **{f"std_{i}": p[1] for i, p in enumerate(preds)})
    if out_path:
        np.savez_compressed(out_path, **res)
    if verbose:
        print(f"{len(kept)} ships fitted ({restarts} restarts each) and predicted in {t1 - t0:.1f} s -> {out_path}")
    return res


def main():
    csv = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "data", "historical_ship_data.csv.gz")
    out_path = sys.argv[2] if len(sys.argv) > 2 else "results_gp_batch.npz"
    restarts = int(sys.argv[3]) if len(sys.argv) > 3 else 50
    run(csv, out_path, restarts=restarts)


if __name__ == "__main__":
    main()
