"""Host-side calendar arithmetic: resample segments, day-of-year tables, expected counts.

All of this is O(T) coordinate work that stays on the host (SURVEY.md §2.1 "Rest of calendar: OUT OF SCOPE"); the
device kernels only see integer tables (``seg_off``, ``tidx``, ``tbase``).  Mirrors the grouping semantics of
``xarray.DataArray.resample(time=freq)`` for the start-anchored frequencies xclim uses (``YS[-MMM]``, ``QS[-MMM]``,
``MS``) and ``core/missing.py:64-160`` (``expected_count``) for the three calendars the survey's fixtures need.
cftime is not installed, so non-standard calendars are handled with integer arithmetic.
"""

from __future__ import annotations

import numpy as np

MONTHS = ["JAN", "FEB", "MAR", "APR", "MAY", "JUN", "JUL", "AUG", "SEP", "OCT", "NOV", "DEC"]
_MLEN_NOLEAP = np.array([31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31])


def _is_leap(year, calendar: str):
    year = np.asarray(year)
    if calendar in ("noleap", "365_day", "360_day"):
        return np.zeros(year.shape, dtype=bool)
    if calendar in ("all_leap", "366_day"):
        return np.ones(year.shape, dtype=bool)
    return ((year % 4 == 0) & (year % 100 != 0)) | (year % 400 == 0)


def _month_len(year: int, month: int, calendar: str) -> int:
    if calendar == "360_day":
        return 30
    n = int(_MLEN_NOLEAP[month - 1])
    if month == 2 and bool(_is_leap(year, calendar)):
        n += 1
    return n


_WEEKDAYS = ["MON", "TUE", "WED", "THU", "FRI", "SAT", "SUN"]


def parse_freq(freq: str) -> tuple[str, int]:
    """Return (base, parameter): base "Y" / "Q" / "M" with the anchor month (start-anchored; the end-anchored spellings
    give the same segments), "W" with the weekday the week ENDS on (0 = Monday; pandas "W" == "W-SUN": bins closed and
    labelled on the right), "D" with the number of days per bin ("D", "7D", "10D": bins from the first day on)."""
    f = freq.upper()
    if f == "W" or f.startswith("W-"):
        day = f.split("-", 1)[1] if "-" in f else "SUN"
        if day not in _WEEKDAYS:
            raise ValueError(f"Unknown anchor weekday in frequency {freq!r}")
        return "W", _WEEKDAYS.index(day)
    if f.endswith("D") and (f[:-1].isdigit() or f == "D"):
        n = int(f[:-1]) if f[:-1] else 1
        if n < 1:
            raise ValueError(f"Bad day count in frequency {freq!r}")
        return "D", n
    for old, new in (("AS", "YS"), ("A-", "Y-")):
        if f.startswith(old):
            f = new + f[len(old):]
    if f in ("MS", "ME", "M"):
        return "M", 1  # end-anchored offsets give the same segments (only the labels differ)
    for end, start in (("YE", "YS"), ("QE", "QS"), ("Y", "YS"), ("Q", "QS")):
        # "YE-MMM"/"QE-MMM": periods END in month MMM -> they START the month after
        if f == end:
            f = start  # year/quarter ending in DEC == starting in JAN
        elif f.startswith(end + "-") and not f.startswith(start):
            mon = f.split("-", 1)[1]
            if mon not in MONTHS:
                raise ValueError(f"Unknown anchor month in frequency {freq!r}")
            f = f"{start}-{MONTHS[(MONTHS.index(mon) + 1) % 12]}"
    for base in ("YS", "QS"):
        if f == base:
            return base[0], 1
        if f.startswith(base + "-"):
            mon = f.split("-", 1)[1]
            if mon not in MONTHS:
                raise ValueError(f"Unknown anchor month in frequency {freq!r}")
            return base[0], MONTHS.index(mon) + 1
    raise NotImplementedError(f"Resampling frequency {freq!r} is not supported by the HIP backend host helper.")


class TimeAxis:
    """A daily (or coarser) time coordinate reduced to integer fields."""

    def __init__(self, year, month, day, calendar: str = "standard"):
        self.year = np.asarray(year, dtype=np.int64)
        self.month = np.asarray(month, dtype=np.int64)
        self.day = np.asarray(day, dtype=np.int64)
        self.calendar = calendar
        if calendar == "360_day":
            self.doy = (self.month - 1) * 30 + self.day
        else:
            cum = np.concatenate([[0], np.cumsum(_MLEN_NOLEAP)])[:-1]
            self.doy = cum[self.month - 1] + self.day + ((self.month > 2) & _is_leap(self.year, calendar))
        self.doy = self.doy.astype(np.int64)

    def __len__(self):
        return int(self.year.shape[0])

    # ---- constructors ----
    @classmethod
    def from_pandas(cls, index) -> "TimeAxis":
        return cls(index.year.values, index.month.values, index.day.values, "standard")

    @classmethod
    def daily(cls, start: str, periods: int, calendar: str = "standard") -> "TimeAxis":
        """Daily axis starting at ``YYYY-MM-DD``."""
        y, m, d = (int(p) for p in start.split("-"))
        if calendar == "standard":
            import pandas as pd

            return cls.from_pandas(pd.date_range(start, periods=periods, freq="D"))
        years = np.empty(periods, dtype=np.int64)
        months = np.empty(periods, dtype=np.int64)
        days = np.empty(periods, dtype=np.int64)
        i = 0
        while i < periods:
            ml = _month_len(y, m, calendar)
            n = min(ml - d + 1, periods - i)
            years[i : i + n] = y
            months[i : i + n] = m
            days[i : i + n] = np.arange(d, d + n)
            i += n
            d = 1
            m += 1
            if m > 12:
                m = 1
                y += 1
        return cls(years, months, days, calendar)

    def days_in_month(self) -> np.ndarray:
        """Length of every step's month (``DatetimeIndex.days_in_month``; cftime's ``daysinmonth``)."""
        if self.calendar == "360_day":
            return np.full(len(self), 30, dtype=np.int64)
        n = _MLEN_NOLEAP[self.month - 1].astype(np.int64)
        return n + ((self.month == 2) & _is_leap(self.year, self.calendar))

    def dates(self) -> np.ndarray:
        """The time coordinate itself (what ``coord=True`` returns in the reference, rl:586-593 -> utils.lazy_indexing):
        ``datetime64[ns]`` when every date of the calendar is a Gregorian date (standard, proleptic_gregorian, noleap);
        for all_leap / 360_day calendars — whose Feb 29 / Feb 30 have no datetime64 — ISO ``YYYY-MM-DD`` strings (cftime,
        the reference's type for them, is not a dependency)."""
        iso = np.array([f"{y:04d}-{m:02d}-{d:02d}" for y, m, d in zip(self.year, self.month, self.day)])
        if self.calendar in ("standard", "gregorian", "proleptic_gregorian", "noleap", "365_day"):
            return iso.astype("datetime64[ns]")
        return iso.astype(object)

    def subset(self, sl) -> "TimeAxis":
        out = TimeAxis.__new__(TimeAxis)
        out.year, out.month, out.day, out.doy = self.year[sl], self.month[sl], self.day[sl], self.doy[sl]
        out.calendar = self.calendar
        return out

    def ordinal(self) -> np.ndarray:
        """Day number of every time step in its own calendar (consecutive days differ by 1); for the Gregorian calendars
        it is the proleptic Gregorian ordinal (0001-01-01 = 1, a Monday)."""
        y, m, d = self.year, self.month, self.day
        if self.calendar == "360_day":
            return y * 360 + (m - 1) * 30 + d
        cum = np.concatenate([[0], np.cumsum(_MLEN_NOLEAP)])[:-1]
        if self.calendar in ("noleap", "365_day"):
            return y * 365 + cum[m - 1] + d
        if self.calendar in ("all_leap", "366_day"):
            return y * 366 + cum[m - 1] + d + (m > 2)
        y1 = y - 1
        return y1 * 365 + y1 // 4 - y1 // 100 + y1 // 400 + self.doy

    # ---- resample segments ----
    def _day_segments(self, base: str, par: int):
        """Weekly / N-day bins: (seg_off, starts) with ``starts`` = (year, month, day) of each bin's first day."""
        o = self.ordinal()
        if np.any(np.diff(o) < 0):
            raise ValueError("time axis must be sorted")
        if base == "W":
            if self.calendar not in ("standard", "gregorian", "proleptic_gregorian"):
                raise NotImplementedError("weekly resampling needs a Gregorian calendar (weekdays)")
            first = (par + 1) % 7                      # weekday the bins start on (Monday = 0; ordinal 1 is a Monday)
            key = (o - 1 - first) // 7
            nd, start_of = 7, lambda k: k * 7 + 1 + first
        else:
            key = (o - o[0]) // par
            nd, start_of = par, lambda k: int(o[0]) + k * par
        k0, k1 = int(key[0]), int(key[-1])
        keys = np.arange(k0, k1 + 1)
        seg_off = np.searchsorted(key, np.concatenate([keys, [k1 + 1]]), side="left").astype(np.int64)
        # labels: walk from a known (ordinal, date) pair; bins are short, so the first step inside / after the bin start
        starts = []
        for k in keys:
            s0 = start_of(int(k))
            i = int(np.searchsorted(o, s0, side="left"))
            if i < len(o) and int(o[i]) - s0 < nd:
                back = int(o[i]) - s0            # days between the bin start and its first present step
                yy, mm, dd = int(self.year[i]), int(self.month[i]), int(self.day[i])
                while back > 0:                  # step the date back inside the calendar
                    dd -= 1
                    if dd < 1:
                        mm -= 1
                        if mm < 1:
                            mm, yy = 12, yy - 1
                        dd = _month_len(yy, mm, self.calendar)
                    back -= 1
                starts.append((yy, mm, dd))
            else:
                starts.append(None)              # an empty bin inside the span
        return seg_off, starts, nd

    def _period_key(self, freq: str):
        base, anchor = parse_freq(freq)
        m0 = self.year * 12 + (self.month - 1)
        if base == "M":
            return m0, 1, 0
        if base == "Y":
            off = anchor - 1
            return (m0 - off) // 12, 12, off
        off = (anchor - 1) % 3
        return (m0 - off) // 3, 3, off

    def segments(self, freq: str):
        """Period segments of ``resample(time=freq)``.

        Returns ``(seg_off, starts)``: ``seg_off`` int64[P+1] offsets into the (sorted) time axis — empty periods
        inside the span are kept, as pandas does — and ``starts`` a list of ``(year, month)`` period start labels.
        """
        base0, par0 = parse_freq(freq)
        if len(self) == 0:
            return np.zeros(1, dtype=np.int64), []
        if base0 in ("W", "D"):
            seg_off, starts, _ = self._day_segments(base0, par0)
            return seg_off, starts
        key, nmon, off = self._period_key(freq)
        if len(key) == 0:
            return np.zeros(1, dtype=np.int64), []
        if np.any(np.diff(key) < 0):
            raise ValueError("time axis must be sorted")
        k0, k1 = int(key[0]), int(key[-1])
        keys = np.arange(k0, k1 + 1)
        seg_off = np.searchsorted(key, np.concatenate([keys, [k1 + 1]]), side="left").astype(np.int64)
        starts = []
        for k in keys:
            mstart = int(k) * nmon + off
            starts.append((mstart // 12, mstart % 12 + 1))
        return seg_off, starts

    def expected_count(self, freq: str, **indexer) -> np.ndarray:
        """Days in each full resampling period (core/missing.py:137-159, daily source).  With a time selection
        (``season=`` / ``month=`` / ``doy_bounds=`` / ``date_bounds=``, core/missing.py:118-135): the selected days of a
        complete synthetic series covering the same periods."""
        if indexer and any(v is not None for k, v in indexer.items() if k != "include_bounds"):
            from .calendar import select_time_mask

            full = self.expected_count(freq)
            _, starts = self.segments(freq)
            y0, m0 = starts[0]
            synth = TimeAxis.daily(f"{y0:04d}-{m0:02d}-01", int(full.sum()), self.calendar)
            mask = select_time_mask(synth, **indexer)
            edges = np.concatenate(([0], np.cumsum(full)))
            return np.array([int(mask[a:b].sum()) for a, b in zip(edges[:-1], edges[1:])], dtype=np.int32)
        base, par = parse_freq(freq)
        if base in ("W", "D"):
            seg_off, _, nd = self._day_segments(base, par)
            return np.full(len(seg_off) - 1, nd, dtype=np.int32)
        _, starts = self.segments(freq)
        nmon = {"M": 1, "Q": 3, "Y": 12}[base]
        out = np.zeros(len(starts), dtype=np.int32)
        for i, (y, m) in enumerate(starts):
            n = 0
            for _ in range(nmon):
                n += _month_len(y, m, self.calendar)
                m += 1
                if m > 12:
                    m, y = 1, y + 1
            out[i] = n
        return out

    # ---- day-of-year tables ----
    def doy_table(self):
        """(tbase int32[nyears, ndoy], years, doys): time index of each (year, doy), -1 when absent (cal:450-458)."""
        years = np.unique(self.year)
        doys = np.unique(self.doy)
        tb = np.full((len(years), len(doys)), -1, dtype=np.int32)
        yi = np.searchsorted(years, self.year)
        di = np.searchsorted(doys, self.doy)
        tb[yi, di] = np.arange(len(self), dtype=np.int32)
        return tb, years, doys

    def max_doy(self) -> int:
        return {"360_day": 360, "noleap": 365, "365_day": 365}.get(self.calendar, 366)
