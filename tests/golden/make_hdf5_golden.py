"""Generator and checker for the HDF5 / netCDF-4 writer (lisflood_amd.hdf5_min), to be run with an interpreter that has
h5py (this image: /opt/conda/bin/python3.9 -- the product interpreter has no HDF5 library).

  make    tests/golden/h5py_earliest.h5: a small file written BY libhdf5 (h5py, libver='earliest') with the structures of
          a LISFLOOD map stack (dimension scales x, y, time; a chunked, shuffled, deflated [T, H, W] variable with a fill
          value; string / int / float attributes).  The CPU tests read it with lisflood_amd.hdf5_min.read: an
          independent pin of the reader, and through it of the writer.
  check   writes a map stack with lisflood_amd.output.write_netcdf4 and opens it with h5py: shapes, values, chunking,
          filters, fill value, attributes, attached dimension scales, REFERENCE_LIST / DIMENSION_LIST targets.  The log
          goes to stdout (committed as profiles/r02_netcdf4_h5py_check.txt)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "lisflood-code_amd"))


def make():
    import h5py
    f = h5py.File(os.path.join(HERE, "h5py_earliest.h5"), "w", libver="earliest")
    f.attrs.create("Conventions", np.string_("CF-1.6"))
    f.attrs.create("source", np.string_("Lisflood output maps"))
    x = f.create_dataset("x", data=np.arange(7, dtype="f8") * 5000.0 + 2502500.0)
    y = f.create_dataset("y", data=(np.arange(5, dtype="f8") * 5000.0 + 752500.0)[::-1].copy())
    t = f.create_dataset("time", data=np.arange(70, dtype="f8"))
    for n, ds, i in (("x", x, 0), ("y", y, 1), ("time", t, 2)):
        ds.make_scale(n)
        ds.attrs.create("_Netcdf4Dimid", np.int32(i))
    t.attrs.create("units", np.string_("days since 2016-01-02 06:00:00.0"))
    rng = np.random.default_rng(11)
    a = rng.uniform(0.0, 250.0, (70, 5, 7))
    a[:, 0, :2] = -9999.0
    d = f.create_dataset("dis", shape=a.shape, dtype="f8", chunks=(1, 5, 7), compression="gzip", compression_opts=4,
                         shuffle=True, fillvalue=-9999.0)
    d[...] = a
    d.dims[0].attach_scale(t); d.dims[1].attach_scale(y); d.dims[2].attach_scale(x)
    d.attrs.create("units", np.string_("m3/s"))
    d.attrs.create("_FillValue", np.array([-9999.0]))
    f.create_dataset("wide", data=rng.uniform(0, 1, (9, 11)).astype("f4"), chunks=(4, 4), compression="gzip")  # edge chunks
    f.create_dataset("code", data=np.arange(6, dtype="i4"))
    f.close()


def make_sparse():
    """tests/golden/h5py_sparse.h5: a chunked variable with a fill value of which libhdf5 allocated only SOME chunks
    (incremental allocation: steps 1 and 3 of 4 were never written) and one of which nothing was written at all"""
    import h5py
    f = h5py.File(os.path.join(HERE, "h5py_sparse.h5"), "w", libver="earliest")
    d = f.create_dataset("state", shape=(4, 3, 5), dtype="f8", chunks=(1, 3, 5), fillvalue=-9999.0, compression="gzip")
    d[0] = np.arange(15, dtype="f8").reshape(3, 5)
    d[2] = np.arange(15, dtype="f8").reshape(3, 5) * 2.0
    f.create_dataset("never", shape=(2, 3), dtype="f4", chunks=(1, 3), fillvalue=-1.5)
    f.close()


def check():
    import tempfile
    import h5py
    from lisflood_amd import output as O
    H, W, T = 57, 80, 12
    rng = np.random.default_rng(5)
    maps = rng.uniform(0.0, 900.0, (T, H, W))
    maps[:, rng.uniform(size=(H, W)) < 0.4] = np.nan
    x = np.arange(W) * 5000.0 + 4322500.0
    y = (np.arange(H) * 5000.0 + 2402500.0)[::-1]
    path = os.path.join(tempfile.mkdtemp(), "dis.nc")
    O.write_netcdf4(path, "dis", maps, x, y, time_values=np.arange(T, dtype=float), time_units="days since 2016-01-02 06:00:00.0",
                    standard_name="DischargeMaps", long_name="ChanQAvg", units="m3/s",
                    coord_attrs={"x": {"standard_name": "projection_x_coordinate", "units": "Meter"},
                                 "y": {"standard_name": "projection_y_coordinate", "units": "Meter"}},
                    projection=("laea", {"grid_mapping_name": "lambert_azimuthal_equal_area", "false_easting": 4321000.0}),
                    esri_pe_string='PROJCS["ETRS_1989_LAEA"]')
    f = h5py.File(path, "r")
    d = f["dis"]
    print("file size", os.path.getsize(path), "bytes; objects", sorted(f.keys()))
    print("dis", d.shape, d.dtype, "chunks", d.chunks, d.compression, d.compression_opts, "shuffle", d.shuffle, "fill", d.fillvalue)
    got = d[...]
    want = np.where(np.isnan(maps), -9999.0, maps)
    print("values identical:", bool(np.array_equal(got, want)))
    print("coordinates identical:", bool(np.array_equal(f["x"][...], x) and np.array_equal(f["y"][...], y)))
    print("scales attached:", [[s.name for s in dim.values()] for dim in d.dims])
    print("DIMENSION_LIST ->", [f[r[0]].name for r in d.attrs["DIMENSION_LIST"]])
    for n in ("x", "y", "time"):
        print(n, "is_scale", bool(h5py.h5ds.is_scale(f[n].id)), "NAME", f[n].attrs["NAME"], "dimid", int(f[n].attrs["_Netcdf4Dimid"]),
              "REFERENCE_LIST ->", [(f[r[0]].name, int(r[1])) for r in f[n].attrs["REFERENCE_LIST"]])
    print("dis attrs", {k: (v if not isinstance(v, np.ndarray) else v.tolist()) for k, v in d.attrs.items() if k != "DIMENSION_LIST"})
    print("time attrs", {k: v for k, v in f["time"].attrs.items() if k not in ("REFERENCE_LIST",)})
    print("laea", f["laea"].shape, f["laea"][()], dict(f["laea"].attrs))
    print("root attrs", {k: v for k, v in f.attrs.items() if k != "date_created"})
    assert np.array_equal(got, want)
    # a long series: 400 chunks in ONE chunk B-tree node (the writer raises the superblock's indexed-storage K), float32
    T2 = 400
    m2 = rng.uniform(0.0, 5.0, (T2, 9, 13))
    m2[:, 0, 0] = np.nan
    path2 = os.path.join(os.path.dirname(path), "long.nc")
    O.write_netcdf4(path2, "dis", m2, np.arange(13) * 1.0, np.arange(9)[::-1] * 1.0, time_values=np.arange(T2) * 1.0, dtype="f4")
    d2 = h5py.File(path2, "r")["dis"]
    ok = bool(np.array_equal(d2[...], np.where(np.isnan(m2), -9999.0, m2).astype("f4")))
    print("long series", d2.shape, d2.dtype, "chunks", d2.chunks, d2.compression, "values identical:", ok)
    assert ok
    # the streamed writer: steps appended out of order, two of them never written (libhdf5 returns the fill value there),
    # the file opened by libhdf5 after a flush() in the middle and after close()
    path3 = os.path.join(os.path.dirname(path), "stream.nc")
    w = O.NetCDF4MapWriter(path3, "dis", x, y, time_values=np.arange(T, dtype=float), units="m3/s")
    order = [3, 0, 1, 7, 2, 11, 5, 6, 4, 9]
    want3 = np.full((T, H, W), -9999.0)
    for i, t in enumerate(order):
        w.write_step(t, maps[t])
        want3[t] = want[t]
        if i == 4:
            w.flush()
            part = h5py.File(path3, "r")["dis"][...]
            okp = bool(np.array_equal(part[[3, 0, 1, 7, 2]], want[[3, 0, 1, 7, 2]]) and (part[[4, 5, 6, 8, 9, 10, 11]] == -9999.0).all())
            print("streamed, after flush() at 5 of 12 steps: libhdf5 reads the written steps and the fill value elsewhere:", okp)
            assert okp
    w.close()
    f3 = h5py.File(path3, "r")
    ok3 = bool(np.array_equal(f3["dis"][...], want3))
    print("streamed", f3["dis"].shape, "10 of 12 steps written out of order: values identical, unwritten steps = fill:", ok3,
          "| scales attached:", [[s.name for s in dim.values()] for dim in f3["dis"].dims])
    assert ok3


def refstruct():
    """The STRUCTURE (no data) of the reference's own dis.nc -- variables, shapes, types, chunking, filters, fill values,
    attribute names and the small attribute values -- as tests/golden/ref_disnc_structure.json: what `writenet` really
    produces (netcdf.py:432-583), for the CPU test that compares write_netcdf4's files with it."""
    import json
    import h5py
    src = "/root/reference/tests/data/LF_ETRS89_UseCase/reference/output_reference_daily/dis.nc"
    f = h5py.File(src, "r")
    skip = {"DIMENSION_LIST", "REFERENCE_LIST", "_Netcdf4Coordinates", "_Netcdf4Dimid", "CLASS", "NAME", "_NCProperties"}

    def attrs(o):
        out = {}
        for k, v in o.attrs.items():
            if k in skip:
                continue
            if isinstance(v, bytes):
                out[k] = v.decode()
            elif np.ndim(v) == 0:
                out[k] = float(v) if np.asarray(v).dtype.kind == "f" else int(v)
            else:
                out[k] = np.asarray(v).tolist()
        return out
    out = {"source": "tests/data/LF_ETRS89_UseCase/reference/output_reference_daily/dis.nc", "root_attrs": sorted(attrs(f)),
           "variables": {}}
    for name, d in f.items():
        out["variables"][name] = dict(shape=list(d.shape), dtype=str(d.dtype), chunks=list(d.chunks) if d.chunks else None,
                                      compression=d.compression, compression_opts=d.compression_opts, shuffle=bool(d.shuffle),
                                      fillvalue=float(d.fillvalue), dims=[s[0].name.strip("/") if len(s) else "" for s in d.dims],
                                      attrs=attrs(d))
    json.dump(out, open(os.path.join(HERE, "ref_disnc_structure.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True)[:1500])


if __name__ == "__main__":
    {"make": make, "sparse": make_sparse, "check": check, "refstruct": refstruct}[sys.argv[1]]()
