#!/usr/bin/env python3
"""
Batch UKF + RTS smoother over every ship of a data file, on one MI355X.

Counterpart of the reference's examples/example_ukf_rts_smoother_batch.py:15-90 with its per-ship Python loop replaced by
``batch.run_fleet``: one batched launch for a file of a hundred ships, chip-sized windows of one resident fleet through the
pipelined kernels for a hundred thousand.  Same ship selection (``ids.pop(1)`` drops the 'id.tidy' pseudo id, :16-17), same matrices (:43-52),
same prior ``x0 = z[:, 0]`` (:60), same 2 sub-steps (:68), same ``dt > 48 h`` skip (:70-72) and the same "error in one
ship does not stop the others" behaviour (:73-90; here: the track's status word).  Headless: instead of the cartopy PDF
it writes one ``.npz`` with the filtered and smoothed histories.

    python examples/example_ukf_rts_smoother_batch.py [historical_ship_data.csv[.gz]] [out.npz]
"""
import os
import sys
import time

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ship-track-estimators_amd"))

from track_estimators import batch  # noqa: E402
from track_estimators.ship_track import ShipTrack  # noqa: E402
from track_estimators.utils import generate_dts  # noqa: E402


def run(csv, out_path=None, sphere=False, seed_base=None, verbose=True):
    """The compute part of the example.  ``sphere``: use the reference's pure-NumPy sphere pair instead of the WGS84
    default (what the test-suite's reference-run fixture was made with: geographiclib is not installed there).
    ``seed_base``: seed NumPy's global generator with ``seed_base + position in ids`` before each ship's noise is
    drawn, so that a run can be replayed (the reference script leaves the generator unseeded).  Returns the result dict
    (also written to ``out_path`` when given) plus ``ids`` and ``category`` (ok / skipped / error) for every id."""
    from track_estimators.utils import haversine_formula, heading

    say = print if verbose else (lambda *a, **k: None)
    df = pd.read_csv(csv)
    ids = df["primary.id"].unique().tolist()
    ids.pop(1)

    H = np.diag([1, 1, 0, 0])
    R = np.diag([0.25, 0.25, 0, 0])
    Q = np.diag([1e-4, 1e-4, 1e-6, 1e-6])
    P = np.diag([1.0, 1.0, 1.0, 1.0])

    t0 = time.perf_counter()
    tracks, dts, kept, kept_pos = [], [], [], []
    category = ["error"] * len(ids)
    for i, sid in enumerate(ids):
        try:
            st = ShipTrack(calc_distance_func=haversine_formula, calc_heading_func=heading) if sphere else ShipTrack()
            st.read_csv(csv_file=csv, ship_id=sid, id_col="primary.id", lat_col="lat", lon_col="lon2", reverse=False)
            if len(st.lon) < 2:
                raise IndexError("fewer than 2 observations")
        except Exception as exc:  # a ship whose rows cannot be parsed does not stop the batch
            say("Error in ", sid, type(exc).__name__)
            continue
        dt_array = generate_dts(st.dts, 2)
        if len(dt_array) == 0 or dt_array.max() > 48:
            say("Skipping becasue dt > 48", sid, dt_array.max() if len(dt_array) else None)
            category[i] = "skipped"
            continue
        tracks.append(st)
        dts.append(dt_array)
        kept.append(sid)
        kept_pos.append(i)
    # sog / cog / rates / z of every kept ship in one launch (the reference: get_measurements + calculate_*_rate per ship)
    batch.prepare_ship_tracks(tracks)
    x0s = [st.z[:, 0].copy() for st in tracks]
    t1 = time.perf_counter()
    noise = []
    for i, d, st in zip(kept_pos, dts, tracks):  # what the reference injects, drawn with its calls in its order
        if seed_base is not None:
            np.random.seed(seed_base + i)
        noise.append(batch.draw_reference_noise(Q, R, d, st.dts))
    hb = batch.pack_tracks(tracks, dts, x0s, H, Q, R, P, noise=noise)
    # the reference's per-ship loop (:19-90) as windows of one resident fleet: one launch for a file of a hundred ships,
    # chip-sized windows through the pipelined kernels for a fleet of a hundred thousand
    out = batch.run_fleet(hb, smooth=True)
    t2 = time.perf_counter()
    for i, sid, s in zip(kept_pos, kept, out["status"]):
        if s & 0x11:  # non-finite state (the reference: LinAlgError / IndexError inside run) -> "Error in", carry on
            say("Error in ", sid)
        else:
            category[i] = "ok"
    res = dict(ids=np.array([str(s) for s in ids]), category=np.array(category), kept=np.array([str(s) for s in kept]),
               kept_pos=np.array(kept_pos), nsteps=out["nsteps"], status=out["status"], means=out["means"],
               covs=out["covs"], means_smoothed=out["means_smoothed"], covs_smoothed=out["covs_smoothed"])
    if out_path:
        np.savez_compressed(out_path, **res)
    say(f"{len(kept)} ships ({int(hb.nsteps.sum())} track-steps), {category.count('error')} failed, "
        f"{category.count('skipped')} skipped; host prep {t1 - t0:.2f} s, pack + GPU + download {t2 - t1:.2f} s -> {out_path}")
    return res


def main():
    csv = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "data", "historical_ship_data.csv.gz")
    out_path = sys.argv[2] if len(sys.argv) > 2 else "results_ukf_rts_batch.npz"
    run(csv, out_path)


if __name__ == "__main__":
    main()
