"""Time-series output in the reference's `.tss` layout (pcrcalc `timeoutput` style, global_modules/zusatz.py:196-290)
and state snapshots of the resident hot path.

A `.tss` file is text: one header line, the number of columns + 1, the word `timestep`, one line per outlet id, then
one row per time step -- the step as ` %8g`, every value as ` %14g`, missing values as `1e31` right-aligned in 15
characters.  The reference samples a map (e.g. ChanQAvg for `dis.tss`, Lisflood_dynamic.py:209 + output.py:565-575)
at the pixels of an id map; `sample()` does that for a compressed vector.
"""
import time

import numpy as np

MISSING = "           1e31"


def sample(vector, pixel_index):
    """values of a compressed [N] vector at the outlet pixels (compressed indices, in id order)"""
    return np.asarray(vector, dtype=np.float64)[np.asarray(pixel_index, dtype=np.int64)]


def header_line(datatype="valuescale.scalar", settings_path="", date=None):
    """zusatz.py:209-211"""
    return "timeseries {} settingsfile: {} date: {}\n".format(datatype.lower(), settings_path,
                                                              time.ctime(time.time()) if date is None else date)


def write_tss(path, ids, first_timestep, values, header=True, datatype="valuescale.scalar", settings_path="", date=None,
              first_line=None):
    """ids: outlet ids (column order); values: [T, len(ids)] (NaN = missing); rows are numbered from first_timestep.
    header=False reproduces the reference's `noheader` flag.  first_line overrides the generated header line."""
    values = np.atleast_2d(np.asarray(values, dtype=np.float64))
    ids = list(ids)
    if values.shape[1] != len(ids):
        raise ValueError("values must have one column per id")
    with open(path, "w") as f:
        if header:
            f.write(first_line if first_line is not None else header_line(datatype, settings_path, date))
            f.write(str(len(ids) + 1) + "\n")
            f.write("timestep\n")
            for c in ids:
                f.write(str(c) + "\n")
        for t in range(values.shape[0]):
            row = " %8g" % (first_timestep + t)
            for x in values[t]:
                row += MISSING if x != x else " %14g" % x
            f.write(row + "\n")


def read_tss(path):
    """-> (first line, ids, first_timestep, values[T, len(ids)])"""
    with open(path) as f:
        first = f.readline()
        ncols = int(f.readline())
        assert f.readline().strip() == "timestep"
        ids = [int(f.readline()) for _ in range(ncols - 1)]
        rows = [ln.split() for ln in f if ln.strip()]
    steps = np.array([int(float(r[0])) for r in rows])
    vals = np.array([[np.nan if x == "1e31" else float(x) for x in r[1:]] for r in rows], dtype=np.float64)
    assert (np.diff(steps) == 1).all()
    return first, ids, int(steps[0]), vals


class TssWriter:
    """Collects one row per model step and writes the file at the end, like TimeoutputTimeseries.sample() /
    _writeTssFile (zusatz.py:246-290)."""

    def __init__(self, path, ids, pixel_index, first_timestep=1, how="", router=None, pixel_area=None, inv_up_area=None,
                 **header_kw):
        """`how`: the `operation` of the reference's time-series definitions (global_modules/output.py:568-575):
        ""            the value at the gauge pixels;
        "mapmaximum"  the maximum of the map at every gauge;
        "total"       catchmenttotal(x * PixelArea, Ldd) * InvUpArea -- the area-weighted mean of x over everything
                      upstream of the gauge.  PCRaster's catchmenttotal is the accumulation over all upstream cells incl.
                      the cell: `router.accuflux` (a kinematicWave or anything with accuflux(x) over the same pixels; the
                      device sweep of lf_accuflux), `pixel_area` and `inv_up_area` as self.var.PixelArea / InvUpArea."""
        if how not in ("", "mapmaximum", "total"):
            raise ValueError("unknown time-series operation %r" % (how,))
        if how == "total" and (router is None or pixel_area is None or inv_up_area is None):
            raise ValueError("how='total' needs router, pixel_area and inv_up_area")
        self.path, self.ids, self.pix, self.first = path, list(ids), np.asarray(pixel_index, np.int64), first_timestep
        self.how, self.router, self.pixel_area, self.inv_up_area = how, router, pixel_area, inv_up_area
        self.rows, self.header_kw = [], header_kw

    def sample(self, vector):
        v = np.asarray(vector, dtype=np.float64)
        if self.how == "mapmaximum":
            v = np.full(v.shape, np.nanmax(v) if v.size else np.nan)
        elif self.how == "total":
            v = self.router.accuflux(v * self.pixel_area) * self.inv_up_area
        self.rows.append(sample(v, self.pix))

    def close(self):
        write_tss(self.path, self.ids, self.first, np.array(self.rows).reshape(len(self.rows), len(self.ids)),
                  **self.header_kw)


# ------------------------------------------------------------------------------------------------------------------
# Maps.  The reference writes netCDF-4 (zlib, chunks (1, H, W), netcdf.py:432-583); the image has no HDF5 writer, so
# the same structure is written as netCDF-3 (classic, 64-bit offsets) through scipy: same dimensions
# (time, y|lat, x|lon), coordinate variables, `_FillValue` = -9999, CF attributes -- files that netCDF4 / xarray
# (and therefore the reference's own map reader) open like the originals; only compression and chunking differ.
# ------------------------------------------------------------------------------------------------------------------
FILL = -9999.0


def decompress(vector, land_mask, fill=FILL):
    """compressed [N] vector -> [H, W] map, `fill` outside the land mask (add1.py:285-305)"""
    land_mask = np.asarray(land_mask, bool)
    out = np.full(land_mask.shape, fill, dtype=np.float64)
    out[land_mask] = np.asarray(vector, dtype=np.float64)
    return out


def write_netcdf_classic(path, var_name, maps, x, y, time_values=None, time_units="days since 1990-01-01 00:00:00.0",
                         calendar="proleptic_gregorian", dtype="f8", standard_name="", long_name="", units="",
                         dims=("y", "x"), settings_path=""):
    """maps: [H, W] (a state / end map) or [T, H, W] with time_values[T] (netcdf.py:552-570); NaN -> _FillValue."""
    from scipy.io import netcdf_file
    maps = np.asarray(maps, dtype=np.float64)
    timed = maps.ndim == 3
    if timed and (time_values is None or len(time_values) != maps.shape[0]):
        raise ValueError("a [T, H, W] stack needs time_values[T]")
    H, W = maps.shape[-2:]
    if len(y) != H or len(x) != W:
        raise ValueError("coordinate vectors do not match the map shape")
    dy, dx = dims
    f = netcdf_file(path, "w", version=2)
    f.settingsfile = settings_path
    f.date_created = time.ctime(time.time())
    f.Source_Software = "lisflood_amd"
    f.source = "Lisflood output maps"
    f.keywords = "Lisflood, EFAS, GLOFAS"
    f.Conventions = "CF-1.6"
    f.createDimension(dx, W)
    vx = f.createVariable(dx, "f8", (dx,)); vx[:] = np.asarray(x, np.float64)
    f.createDimension(dy, H)
    vy = f.createVariable(dy, "f8", (dy,)); vy[:] = np.asarray(y, np.float64)
    if timed:
        f.createDimension("time", maps.shape[0])
        vt = f.createVariable("time", "f8", ("time",))
        vt.standard_name = "time"; vt.calendar = calendar; vt.units = time_units
        vt[:] = np.asarray(time_values, np.float64)
        v = f.createVariable(var_name, dtype, ("time", dy, dx))
    else:
        v = f.createVariable(var_name, dtype, (dy, dx))
    v._FillValue = np.array(FILL, dtype=dtype)
    v.standard_name, v.long_name, v.units = standard_name, long_name, units
    v[:] = np.where(np.isnan(maps), FILL, maps).astype(dtype)
    f.close()


def read_netcdf_classic(path, var_name):
    """-> (maps with NaN at _FillValue, x, y, time or None)"""
    from scipy.io import netcdf_file
    with netcdf_file(path, "r", mmap=False) as f:
        v = f.variables[var_name]
        a = np.array(v[:], dtype=np.float64)
        a[a == float(v._FillValue)] = np.nan
        dy, dx = v.dimensions[-2:]
        t = np.array(f.variables["time"][:]) if "time" in f.variables else None
        return a, np.array(f.variables[dx][:]), np.array(f.variables[dy][:]), t


def write_netcdf4(path, var_name, maps, x, y, time_values=None, time_units="days since 1990-01-01 00:00:00.0",
                  calendar="proleptic_gregorian", dtype="f8", standard_name="", long_name="", units="",
                  dims=("y", "x"), settings_path="", complevel=4, coord_attrs=None, projection=None,
                  esri_pe_string=None):
    """The reference's map writer (netcdf.py:432-583, `writenet`): a netCDF-4 (HDF5) file with dimensions x, y (, time),
    coordinate variables, and the value variable `var_name` -- zlib (deflate `complevel`, shuffle), `_FillValue` -9999,
    chunks (1, H, W) for a [T, H, W] stack (netcdf.py:559), one chunk for a single map.  Written by lisflood_amd.hdf5_min
    (no HDF5 library in the image): old-style HDF5 structures, dimension scales with the netCDF-4 attributes.
    `coord_attrs`: {dimension name: {attribute: value}} for x / y (the reference copies them from its template map),
    `projection`: (variable name, {attributes}) for the scalar int grid-mapping variable (`laea`, netcdf.py:497-504)."""
    maps = np.asarray(maps, dtype=np.float64)
    timed = maps.ndim == 3
    if timed and (time_values is None or len(time_values) != maps.shape[0]):
        raise ValueError("a [T, H, W] stack needs time_values[T]")
    if maps.shape[-2:] != (len(y), len(x)):      # (before the file is created or truncated)
        raise ValueError("coordinate vectors do not match the map shape")
    with NetCDF4MapWriter(path, var_name, x, y, time_values if timed else None, time_units, calendar, dtype, standard_name,
                          long_name, units, dims, settings_path, complevel, coord_attrs, projection, esri_pe_string) as w:
        for t, m in enumerate(maps if timed else [maps]):
            w.write_step(t, m)


class NetCDF4MapWriter:
    """write_netcdf4 one map at a time: the file's metadata (dimensions, coordinate variables, all `time_values`) goes to
    disk when the writer is made, write_step(t, map) appends that step's compressed chunk, close() completes the chunk
    index -- a [T, H, W] output stack never sits in memory (the reference appends to its open netCDF4 dataset the same
    way, netcdf.py:540-583 with `flag_time`).  Steps not written read back as `_FillValue`.  `time_values` None: one map."""

    def __init__(self, path, var_name, x, y, time_values=None, time_units="days since 1990-01-01 00:00:00.0",
                 calendar="proleptic_gregorian", dtype="f8", standard_name="", long_name="", units="", dims=("y", "x"),
                 settings_path="", complevel=4, coord_attrs=None, projection=None, esri_pe_string=None):
        from . import hdf5_min as H5
        timed = time_values is not None
        H, W = len(y), len(x)
        dy, dx = dims
        coord_attrs = coord_attrs or {}
        ds = []
        if projection is not None:
            ds.append(H5.Dataset(projection[0], np.array(0, np.int32), (), dict(projection[1])))
        ds.append(H5.Dataset(dx, np.asarray(x, np.float64), (dx,), coord_attrs.get(dx, {})))
        ds.append(H5.Dataset(dy, np.asarray(y, np.float64), (dy,), coord_attrs.get(dy, {})))
        if timed:
            ds.append(H5.Dataset("time", np.asarray(time_values, np.float64), ("time",),
                                 {"standard_name": "time", "calendar": calendar, "units": time_units}))
        attrs = {"_FillValue": np.array([FILL], dtype=dtype), "standard_name": standard_name, "long_name": long_name,
                 "units": units}
        if esri_pe_string:
            attrs["esri_pe_string"] = esri_pe_string
        T = len(time_values) if timed else 0
        ds.append(H5.Dataset(var_name, None, (("time",) if timed else ()) + (dy, dx), attrs,
                             chunks=((1, H, W) if timed else (H, W)), deflate=complevel, shuffle=True,
                             fill=np.array(FILL, dtype=dtype), shape=((T, H, W) if timed else (H, W)), dtype=dtype))
        self._timed, self._name, self._shape, self._dtype = timed, var_name, (H, W), np.dtype(dtype)
        self._w = H5.Writer(path, ds, {"settingsfile": settings_path, "date_created": time.ctime(time.time()),
                                       "Source_Software": "lisflood_amd", "source": "Lisflood output maps",
                                       "keywords": "Lisflood, EFAS, GLOFAS", "Conventions": "CF-1.6"})

    def write_step(self, t, map2d):
        """`map2d` [H, W] (NaN = missing) as step `t` of the stack (t is ignored for a single map)"""
        m = np.asarray(map2d, dtype=np.float64)
        if m.shape != self._shape:
            raise ValueError("map shape %s, file %s" % (m.shape, self._shape))
        block = np.where(np.isnan(m), FILL, m).astype(self._dtype)
        if self._timed:
            self._w.write_chunk(self._name, (t, 0, 0), block[None])
        else:
            self._w.write_chunk(self._name, (0, 0), block)

    def flush(self):
        """chunk index and end-of-file address as of now: a reader sees every step written so far (call it every few
        steps of a long run; close() -- also on garbage collection -- does it once more)"""
        self._w.flush()

    def close(self):
        self._w.close()

    def __del__(self):
        try:
            self._w.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def read_netcdf4(path, var_name):
    """-> (maps with NaN at _FillValue, x, y, time or None) of a file written by write_netcdf4 (or by libhdf5 with the
    'earliest' format structures)"""
    from . import hdf5_min as H5
    r = H5.read(path)
    a = r.dataset(var_name).astype(np.float64)
    fv = r.attrs(var_name).get("_FillValue")
    if fv is not None:
        a[a == float(np.asarray(fv).reshape(-1)[0])] = np.nan
    names = set(r.objects)
    dx = "x" if "x" in names else "lon"
    dy = "y" if "y" in names else "lat"
    t = r.dataset("time") if "time" in names else None
    return a, r.dataset(dx), r.dataset(dy), t


def write_state_maps(directory, maps, land_mask, x=None, y=None, time_value=None, fmt="netcdf4", **kw):
    """One file per state map, named as the reference names it (`ChanQState.nc`, `Theta1ForestState.nc`, ...: the
    'repStateMaps' entries of default_options.py), each a [1, H, W] stack at the state step -- what a warm run reads back
    through its *InitValue bindings.  `maps`: name -> compressed [N] vector (HotPathDevice.state_maps()).
    `fmt`: "netcdf4" (as the reference writes them) or "classic" (netCDF-3)."""
    import os
    writer = {"netcdf4": write_netcdf4, "classic": write_netcdf_classic}[fmt]
    land_mask = np.asarray(land_mask, bool)
    H, W = land_mask.shape
    x = np.arange(W, dtype=np.float64) if x is None else x
    y = np.arange(H, dtype=np.float64)[::-1] if y is None else y
    os.makedirs(directory, exist_ok=True)
    for name, vec in maps.items():
        stack = decompress(vec, land_mask, fill=np.nan)[None]
        writer(os.path.join(directory, name + ".nc"), name, stack, x, y,
               time_values=[0.0 if time_value is None else float(time_value)], **kw)


def read_state_maps(directory, land_mask, names=None):
    """-> name -> compressed [N] vector; a missing value inside the land mask reads as -9999 (cold start)"""
    import glob
    import os
    land_mask = np.asarray(land_mask, bool)
    out = {}
    for path in sorted(glob.glob(os.path.join(directory, "*.nc"))):
        name = os.path.splitext(os.path.basename(path))[0]
        if names is not None and name not in names:
            continue
        with open(path, "rb") as f:
            hdf5 = f.read(4) == b"\x89HDF"
        a = (read_netcdf4 if hdf5 else read_netcdf_classic)(path, name)[0]
        v = a[-1][land_mask] if a.ndim == 3 else a[land_mask]
        out[name] = np.where(np.isnan(v), FILL, v)
    return out
