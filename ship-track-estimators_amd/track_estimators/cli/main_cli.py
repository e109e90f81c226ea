"""
``track_estimator`` console entry point.  Same inputs, settings keys and output files as the reference CLI
(/root/reference/src/track_estimators/cli/main_cli.py:54-258): ``input.json`` with ``dim, dt, nsteps, H, Q, R, P`` and
optional ``smooth``; outputs ``{prefix}_{id}_predictions.txt``, ``_variances.txt`` (covariance diagonals),
``_dts.txt``, ``original_{id}_track.txt`` and, with ``-rts``, ``_predictions_smoothed.txt`` / ``_variances_smoothed.txt``,
all written with ``np.savetxt``.  The filter and smoother run on the GPU; several comma-separated ship ids are packed
into one batched launch.

Opt-in keys of ``input.json`` beyond the reference's (absent = the reference's behaviour):
  ``"robust": true``      Mahalanobis outlier rejection in every update -- the reference's ``check_robustness``
                          (kalman_filters/unscented.py:353-387), whose call site it ships commented out (:228); BASELINE.json
                          configs[3] runs with it on
  ``"chi_alpha": 50.0``   its threshold (the reference hard-codes 50, unscented.py:357)
  ``"geodesy": "sphere"`` speed / course over ground from ``haversine_formula`` / ``heading`` (utils.py:75-147) instead of
                          the WGS84 pair ``ShipTrack`` defaults to
  ``"drop_duplicate_times": true``  rows whose (hour-resolution) timestamp repeats the row before are dropped on reading
                          (``ShipTrack.read_csv(drop_duplicate_times=True)``): five of the seven data/modern_ships ids carry
                          such rows and end non-finite without it, in the reference (LinAlgError) as here (status NaN)
"""
from __future__ import annotations

import logging
import os
import sys
from typing import Tuple

import numpy as np

from ..kalman_filters.non_linear_process import geodetic_dynamics
from ..kalman_filters.unscented import UnscentedKalmanFilter
from ..ship_track import ShipTrack
from ..utils import generate_dts, geographiclib_distance, geographiclib_heading, haversine_formula, heading, smooth
from .argument_parser import __version__, create_parser
from .json_loader import load_input_json

logger = logging.getLogger(__name__)

_BANNER = r"""
             /|~~~
           ///|
         /////|
       ///////|
     /////////|
   \==========|===/
~~~~~~~~~~~~~~~~~~~~~
"""


def start_banner():
    logger.info(_BANNER, extra={"simple": True})
    logger.info(f"version: {__version__}", extra={"simple": True})


def exit_banner():
    logger.info("Track estimator has terminated succesfully! :)", extra={"simple": True})


def _get_input_matrix(settings: dict, matrix_name: str, dim: int) -> np.ndarray:
    """1-D entries become a diagonal matrix, 2-D entries are taken as they are (main_cli.py:222-258)."""
    if matrix_name not in settings:
        raise KeyError(f"{matrix_name} not found in input settings")
    matrix = np.asarray(settings[matrix_name])
    assert matrix.shape[0] == dim, f"Dimension mismatch: {matrix.shape[0]} != {dim} for {matrix_name}"
    if matrix.ndim == 1:
        return np.diag(matrix)
    if matrix.ndim == 2:
        assert matrix.shape[1] == dim, f"Dimension mismatch: {matrix.shape[1]} != {dim} for {matrix_name}"
        return matrix
    raise ValueError(f"{matrix_name} must be 1 or 2 dimensional")


def get_input_settings(settings: dict) -> Tuple:
    """(dim, dt, nsteps, H, Q, R, P, smooth_control); ``dim``, ``dt``, ``nsteps`` are mandatory (main_cli.py:172-219)."""
    for key in ("dim", "dt", "nsteps"):
        if key not in settings:
            raise KeyError(f"{key} not found in input settings")
    dim = int(settings["dim"])
    dt = settings["dt"]
    nsteps = int(settings["nsteps"])
    smooth_control = int(settings["smooth"]) if "smooth" in settings else None
    H, Q, R, P = (_get_input_matrix(settings, name, dim) for name in ("H", "Q", "R", "P"))
    return dim, dt, nsteps, H, Q, R, P, smooth_control


def get_optional_settings(settings: dict) -> Tuple[bool, float, str, bool]:
    """(robust, chi_alpha, geodesy, drop_duplicate_times): the opt-in keys this CLI adds to the reference's input.json
    (module docstring)."""
    robust = settings.get("robust", False)
    if not isinstance(robust, bool):
        raise ValueError(f"'robust' must be true or false, got {robust!r}")
    chi_alpha = float(settings.get("chi_alpha", 50.0))
    if not chi_alpha > 0.0:
        raise ValueError(f"'chi_alpha' must be positive, got {chi_alpha!r}")
    geodesy = settings.get("geodesy", "wgs84")
    if geodesy not in ("wgs84", "sphere"):
        raise ValueError(f"'geodesy' must be \"wgs84\" or \"sphere\", got {geodesy!r}")
    dedup = settings.get("drop_duplicate_times", False)
    if not isinstance(dedup, bool):
        raise ValueError(f"'drop_duplicate_times' must be true or false, got {dedup!r}")
    return robust, chi_alpha, geodesy, dedup


def _prepare_track(args, ship_id, smooth_control, geodesy="wgs84", dedup=False):
    """ShipTrack -> measurements, rates and prior exactly as main_cli.py:89-109 does."""
    if geodesy == "sphere":
        ship_track = ShipTrack(calc_distance_func=haversine_formula, calc_heading_func=heading)
    else:
        ship_track = ShipTrack(calc_distance_func=geographiclib_distance, calc_heading_func=geographiclib_heading)
    ship_track.read_csv(args.track_file, ship_id=ship_id, id_col=args.id_col, lat_col=args.lat_id, lon_col=args.lon_id,
                        reverse=bool(args.reverse), drop_duplicate_times=dedup)
    if smooth_control not in [-1, 0, 1, None]:
        logger.info(f"Smoothing SOG and COG by {smooth_control}.")
        ship_track.calculate_cog()
        ship_track.calculate_sog()
        ship_track.sog = smooth(ship_track.sog, smooth_control)
        ship_track.cog = smooth(ship_track.cog, smooth_control)
    z = ship_track.get_measurements(include_sog=True, include_cog=True)
    ship_track.calculate_cog_rate()
    ship_track.calculate_sog_rate()
    return ship_track, z[:, 0].reshape(-1, 1).copy()


def _write_outputs(prefix, ship_id, ship_track, dt_array, predictions, estimate_vars, smoothed=None):
    np.savetxt(f"{prefix}_{ship_id}_predictions.txt", np.asarray(predictions))
    np.savetxt(f"{prefix}_{ship_id}_variances.txt", np.diagonal(np.asarray(estimate_vars), axis1=1, axis2=2))
    np.savetxt(f"{prefix}_{ship_id}_dts.txt", np.asarray(dt_array))
    np.savetxt(f"original_{ship_id}_track.txt", np.array((ship_track.lon, ship_track.lat)).T)
    if smoothed is not None:
        np.savetxt(f"{prefix}_{ship_id}_predictions_smoothed.txt", np.asarray(smoothed[0]))
        np.savetxt(f"{prefix}_{ship_id}_variances_smoothed.txt", np.diagonal(np.asarray(smoothed[1]), axis1=1, axis2=2))


def track_estimator(argv=None):
    """Run the track estimator."""
    logging.basicConfig(format="Track estimator | %(levelname)s | %(asctime)s | %(message)s", level=logging.INFO,
                        datefmt="%Y-%m-%d %H:%M:%S", stream=sys.stdout)
    start_banner()
    args = create_parser().parse_args(argv)
    if not os.path.isfile(args.input_file):
        logger.error(f"Input file '{args.input_file}' does not exist.")
        exit_banner()
        return
    if not os.path.isfile(args.track_file):
        logger.error(f"Track file '{args.track_file}' does not exist.")
        exit_banner()
        return
    logger.info(f"Reading input JSON from '{args.input_file}'...")
    settings = load_input_json(args.input_file)
    dim, dt, nsteps, H, Q, R, P, smooth_control = get_input_settings(settings)
    robust, chi_alpha, geodesy, dedup = get_optional_settings(settings)
    substeps = nsteps if dt in [-1, 0, None] else 1  # a positive constant dt is ignored, like main_cli.py:114-120

    ship_ids = [s for s in str(args.ship_id).split(",") if s] if "," in str(args.ship_id) else [args.ship_id]
    if len(ship_ids) == 1 and not robust:
        ship_track, x0 = _prepare_track(args, ship_ids[0], smooth_control, geodesy, dedup)
        dt_array = generate_dts(ship_track.dts, substeps)
        logger.info("Running the Unscented Kalman Filter.")
        ukf = UnscentedKalmanFilter(H=H, Q=Q, R=R, P=P, x0=x0, non_linear_process=geodetic_dynamics)
        if args.no_noise:
            ukf.inject_noise = False
        predictions, estimate_vars = ukf.run(len(dt_array), dt_array, ship_track)
        logger.info("Finished running the Unscented Kalman Filter.")
        smoothed = ukf.run_rts_smoother(ship_track=ship_track) if args.apply_rts_smoother else None
        logger.info(f"Writing outputs with prefix '{args.output_prefix}'.")
        _write_outputs(args.output_prefix, ship_ids[0], ship_track, dt_array, predictions, estimate_vars, smoothed)
    else:
        from .. import batch

        tracks, x0s, dts = [], [], []
        for sid in ship_ids:
            st, x0 = _prepare_track(args, sid, smooth_control, geodesy, dedup)
            tracks.append(st)
            x0s.append(x0[:, 0])
            dts.append(generate_dts(st.dts, substeps))
        logger.info(f"Running the Unscented Kalman Filter on {len(tracks)} track(s) in one batch"
                    + (f", Mahalanobis outlier rejection on (chi_alpha = {chi_alpha:g})." if robust else "."))
        noise = None
        if not args.no_noise:
            noise = [batch.draw_reference_noise(np.asarray(Q), np.asarray(R), d, st.dts) for d, st in zip(dts, tracks)]
        hb = batch.pack_tracks(tracks, dts, x0s, H, Q, R, np.asarray(P, dtype=np.float64), noise=noise)
        hb.robust, hb.chi_alpha = robust, chi_alpha
        # one launch for a few ships, windows through the pipelined kernels for a fleet (batch.run_fleet)
        out = batch.run_fleet(hb, smooth=args.apply_rts_smoother)
        logger.info(f"Writing outputs with prefix '{args.output_prefix}'.")
        for b, sid in enumerate(ship_ids):
            if out["status"][b] & 0x1:
                logger.error(f"Error in {sid}: non-finite state (the reference raises LinAlgError here); skipped.")
                continue
            n1 = out["nsteps"][b] + 1
            sm = (out["means_smoothed"][b, :n1], out["covs_smoothed"][b, :n1]) if args.apply_rts_smoother else None
            _write_outputs(args.output_prefix, sid, tracks[b], dts[b], out["means"][b, :n1], out["covs"][b, :n1], sm)
    exit_banner()


if __name__ == "__main__":  # pragma: no cover
    track_estimator()
