"""
Per-ship data container.  Mirrors reference ``track_estimators.ship_track.ShipTrack``
(/root/reference/src/track_estimators/ship_track.py:9-392): same constructor, attributes and methods.  This is the
input side of the hot path: it produces the arrays ``z, dts, sog, cog, sog_rate, cog_rate`` that the
batch packer (track_estimators/batch.py) lays out SoA for the HIP kernels.  O(T) host work.
"""
from __future__ import annotations

from typing import Callable, Optional, Union

import numpy as np
import pandas as pd

from .utils import geographiclib_distance, geographiclib_heading


class ShipTrack:
    """
    One ship's observations.

    Attributes (ship_track.py:70-87): ``lat, lon, cog, sog, dts, dates, df, sog_rate, cog_rate, z`` and the injected
    ``calc_distance_func(lon1, lat1, lon2, lat2) -> km`` / ``calc_heading_func(...) -> degrees``.
    """

    def __init__(self, csv_file: Optional[str] = None, estimate_cog: bool = False, estimate_sog: bool = False,
                 estimate_sog_rate: bool = False, estimate_cog_rate: bool = False,
                 calc_distance_func: Callable = geographiclib_distance,
                 calc_heading_func: Callable = geographiclib_heading) -> None:
        self.lat = self.lon = self.cog = self.sog = self.dts = self.dates = None
        self.df = None
        self.sog_rate = self.cog_rate = self.z = None
        self.calc_distance_func = calc_distance_func
        self.calc_heading_func = calc_heading_func
        if csv_file is not None:
            self.read_csv(csv_file=csv_file)
            if estimate_sog_rate:
                self.calculate_sog_rate()
            elif estimate_sog:
                self.calculate_sog()
            if estimate_cog_rate:
                self.calculate_cog_rate()
            elif estimate_cog:
                self.calculate_cog()
            self.z = self.get_measurements()

    def read_csv(self, csv_file: str, ship_id: Optional[Union[str, int]] = None, id_col: str = "id",
                 lat_col: str = "lat", lon_col: str = "lon", reverse: bool = False, drop_duplicate_times: bool = False):
        """
        Load the rows of ``ship_id`` from ``csv_file`` (columns ``yr, mo, dy, hr`` + id/lat/lon columns).

        Follows ship_track.py:139-195 including its quirks: the id column and ``ship_id`` are compared as strings
        (``str(None) == 'None'`` filters too), rows keep file order (index sort only), timestamps are built as
        ``yr-mo-dyThh:00:00`` and ``dts`` are the gaps in hours.  Returns ``(lat, lon, dts)``.

        ``drop_duplicate_times`` (an extra of this package, default off = the reference's behaviour): drop every row whose
        timestamp equals that of the row before it.  Hour-resolution timestamps repeat in real logs (five of the seven
        ships of data/modern_ships); a zero gap makes sog = distance / 0 (ship_track.py:217) and the filter ends in NaN /
        LinAlgError.  With the switch on a file reads as the same file with those rows deleted would
        (tests/golden/modern_ships_dedup.npz: the reference run on such files).
        """
        df = pd.read_csv(csv_file)
        df[id_col] = df[id_col].astype(str)
        ship_id = str(ship_id)
        df = df.loc[df[id_col] == ship_id]
        df = df.sort_index(axis=0, ignore_index=True)
        if df.empty:
            raise ValueError(f"No data found for ship '{ship_id}' in '{csv_file}'.")
        stamp = df["yr"].astype(str) + "-" + df["mo"].astype(str) + "-" + df["dy"].astype(str)
        stamp = stamp + "T" + df["hr"].astype(str).str.zfill(2) + ":00:00"
        df = df.assign(date=stamp)
        if drop_duplicate_times:
            when = pd.to_datetime(df.date)
            df = df.loc[(when != when.shift(1)).values].reset_index(drop=True)
        self.df = df
        self.dates = pd.to_datetime(df.date).to_list()
        gaps = [pd.Timedelta(self.dates[i + 1] - self.dates[i]).total_seconds() / 3600.0
                for i in range(len(self.dates) - 1)]
        self.dts = np.asarray(gaps)
        self.lat = pd.to_numeric(df[lat_col]).values
        self.lon = pd.to_numeric(df[lon_col]).values
        assert len(self.lon) > 0, f"Longitude list is empty for column '{lon_col}'."
        assert len(self.lat) > 0, f"Latitude list is empty for column '{lat_col}'."
        assert len(self.lat) == len(self.lon)
        if reverse:
            self.dts, self.lat, self.lon = self.dts[::-1], self.lat[::-1], self.lon[::-1]
        return self.lat, self.lon, self.dts

    def _pairwise(self, fn):
        return [fn(self.lon[i - 1], self.lat[i - 1], self.lon[i], self.lat[i]) for i in range(1, len(self.lon))]

    def calculate_sog(self) -> np.ndarray:
        """Speed over ground per gap = distance / dt, last value repeated (ship_track.py:197-224)."""
        sog = [d / self.dts[i] for i, d in enumerate(self._pairwise(self.calc_distance_func))]
        sog.append(sog[-1])
        self.sog = np.asarray(sog)
        return self.sog

    def calculate_cog(self):
        """Course over ground per gap, last value repeated (ship_track.py:252-278)."""
        cog = self._pairwise(self.calc_heading_func)
        cog.append(cog[-1])
        self.cog = np.asarray(cog)
        return self.cog

    @staticmethod
    def _backward_rate(v, dts):
        rate = [0]
        rate.extend((v[i] - v[i - 1]) / dts[i - 1] for i in range(1, len(v)))
        return np.asarray(rate)

    def calculate_sog_rate(self) -> np.ndarray:
        """Backward difference of sog with a leading 0 (ship_track.py:226-250)."""
        if self.sog is None:
            self.calculate_sog()
        self.sog_rate = self._backward_rate(self.sog, self.dts)
        return self.sog_rate

    def calculate_cog_rate(self) -> np.ndarray:
        """Backward difference of cog with a leading 0 (ship_track.py:280-304)."""
        if self.cog is None:
            self.calculate_cog()
        self.cog_rate = self._backward_rate(self.cog, self.dts)
        return self.cog_rate

    def get_measurements(self, include_sog: bool = False, include_cog: bool = False) -> np.ndarray:
        """Measurement matrix with rows lon, lat[, sog][, cog] (ship_track.py:306-338)."""
        rows = [self.lon, self.lat]
        if include_sog:
            if self.sog is None:
                self.calculate_sog()
            rows.append(self.sog)
        if include_cog:
            if self.cog is None:
                self.calculate_cog()
            rows.append(self.cog)
        self.z = np.vstack(rows)
        return self.z

    def plot_trajectory(self, figsize: tuple = (20, 15), scatter_kwargs: dict = {"s": 10, "color": "red"},
                        savefig: Optional[str] = None, show: bool = True):
        """Scatter the positions on a PlateCarree map (ship_track.py:340-392); needs cartopy + matplotlib."""
        import cartopy.crs as ccrs
        import matplotlib.pyplot as plt

        assert self.lat is not None, "Latitude is not set."
        assert self.lon is not None, "Longitude is not set."
        fig, ax = plt.subplots(nrows=1, ncols=1, subplot_kw={"projection": ccrs.PlateCarree()}, figsize=figsize)
        ax.stock_img()
        ax.coastlines()
        ax.gridlines(crs=ccrs.PlateCarree(), draw_labels=True, linewidth=0.6, color="gray", alpha=0.5, linestyle="-.")
        ax.scatter(self.lon, self.lat, transform=ccrs.PlateCarree(), **scatter_kwargs)
        if savefig:
            plt.savefig(savefig, bbox_inches="tight", dpi=300)
        if show:
            plt.show()
        return fig, ax
