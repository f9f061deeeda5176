"""Oracle time helpers (independent of xclim_amd.timeaxis).  TEST INFRASTRUCTURE ONLY.

Standard calendars go through pandas exactly as xarray's ``resample`` does; noleap / 360_day (cftime is not
installed) use a small explicit (year, month, day) table.
"""

from __future__ import annotations

import numpy as np
import pandas as pd

_ML = [31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31]
_MON = ["JAN", "FEB", "MAR", "APR", "MAY", "JUN", "JUL", "AUG", "SEP", "OCT", "NOV", "DEC"]


class OTime:
    """Daily time coordinate: either a pandas DatetimeIndex (standard) or integer fields (noleap, 360_day)."""

    def __init__(self, index=None, year=None, month=None, day=None, calendar="standard"):
        self.calendar = calendar
        if index is not None:
            self.index = pd.DatetimeIndex(index)
            self.year = self.index.year.values
            self.month = self.index.month.values
            self.day = self.index.day.values
            self.doy = self.index.dayofyear.values
        else:
            self.index = None
            self.year = np.asarray(year)
            self.month = np.asarray(month)
            self.day = np.asarray(day)
            if calendar == "360_day":
                self.doy = (self.month - 1) * 30 + self.day
            else:
                cum = np.cumsum([0] + _ML[:-1])
                self.doy = cum[self.month - 1] + self.day

    def __len__(self):
        return len(self.year)

    @classmethod
    def standard(cls, start, periods):
        return cls(index=pd.date_range(start, periods=periods, freq="D"))

    @classmethod
    def noleap(cls, start_year, periods, calendar="noleap"):
        ml = [30] * 12 if calendar == "360_day" else _ML
        ys, ms, ds = [], [], []
        y, m, d = start_year, 1, 1
        for _ in range(periods):
            ys.append(y), ms.append(m), ds.append(d)
            d += 1
            if d > ml[m - 1]:
                d, m = 1, m + 1
                if m > 12:
                    m, y = 1, y + 1
        return cls(year=ys, month=ms, day=ds, calendar=calendar)

    def isel(self, sl):
        if self.index is not None:
            return OTime(index=self.index[sl])
        return OTime(year=self.year[sl], month=self.month[sl], day=self.day[sl], calendar=self.calendar)


def groups(time: OTime, freq: str):
    """List of (label, index array) for ``resample(time=freq)`` including empty in-span periods."""
    T = len(time)
    if time.index is not None:
        s = pd.Series(np.arange(T), index=time.index)
        out = []
        for label, g in s.resample(freq):
            out.append((label, g.values.astype(np.int64)))
        return out
    f = freq.upper().replace("AS", "YS")
    if f == "MS":
        nmon, off = 1, 0
    elif f.startswith("YS"):
        nmon, off = 12, (_MON.index(f.split("-")[1]) if "-" in f else 0)
    elif f.startswith("QS"):
        nmon, off = 3, ((_MON.index(f.split("-")[1]) if "-" in f else 0) % 3)
    else:
        raise NotImplementedError(freq)
    m0 = time.year * 12 + time.month - 1
    key = (m0 - off) // nmon
    out = []
    for k in range(int(key.min()), int(key.max()) + 1):
        ms = k * nmon + off
        out.append(((ms // 12, ms % 12 + 1), np.nonzero(key == k)[0].astype(np.int64)))
    return out


def days_in_period(time: OTime, freq: str):
    """Expected number of daily steps per full period (reference core/missing.py:137-159)."""
    gs = groups(time, freq)
    out = []
    if time.index is not None:
        labels = pd.DatetimeIndex([g[0] for g in gs])
        ends = labels.shift(1, freq=freq)
        return ((ends - labels) / pd.Timedelta("1D")).astype(int).values
    f = freq.upper().replace("AS", "YS")
    nmon = 1 if f == "MS" else (12 if f.startswith("YS") else 3)
    for (y, m), _ in gs:
        out.append(nmon * 30 if time.calendar == "360_day" else sum(_ML[(m - 1 + i) % 12] for i in range(nmon)))
    return np.array(out)
