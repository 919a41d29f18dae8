"""netCDF-4 (HDF5) map output without an HDF5 library: lisflood_amd.hdf5_min + output.write_netcdf4
(reference: global_modules/netcdf.py:432-583, `writenet`: NETCDF4, zlib, _FillValue -9999, chunks (1, H, W)).

The reader is pinned on a file written by libhdf5 itself (tests/golden/h5py_earliest.h5, made by
tests/golden/make_hdf5_golden.py with h5py); the writer is pinned through the reader here and, in the container, by opening
its files with h5py (`make_hdf5_golden.py check`, log in profiles/r02_netcdf4_h5py_check.txt)."""
import os
import struct

import numpy as np
import pytest

from lisflood_amd import hdf5_min as H5
from lisflood_amd import output as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "h5py_earliest.h5")


def test_reader_on_a_file_written_by_libhdf5():
    r = H5.read(GOLD)
    assert sorted(r.objects) == ["code", "dis", "time", "wide", "x", "y"]
    rng = np.random.default_rng(11)                       # as make_hdf5_golden.py
    a = rng.uniform(0.0, 250.0, (70, 5, 7))
    a[:, 0, :2] = -9999.0
    wide = rng.uniform(0, 1, (9, 11)).astype("f4")
    assert np.array_equal(r.dataset("dis"), a)            # 70 chunks: a two-level chunk B-tree (K = 32), shuffle + deflate
    assert np.array_equal(r.dataset("wide"), wide)        # 3 x 3 chunks of 4 x 4 with edge chunks, float32
    assert np.array_equal(r.dataset("code"), np.arange(6, dtype="i4"))
    assert np.array_equal(r.dataset("x"), np.arange(7) * 5000.0 + 2502500.0)
    assert r.attrs()["Conventions"] == "CF-1.6"
    t = r.attrs("time")
    assert t["CLASS"] == "DIMENSION_SCALE" and t["NAME"] == "time" and int(t["_Netcdf4Dimid"]) == 2
    assert t["units"] == "days since 2016-01-02 06:00:00.0"
    assert float(r.attrs("dis")["_FillValue"][0]) == -9999.0


@pytest.mark.parametrize("dtype", ["f8", "f4"])
def test_map_stack_round_trip(tmp_path, dtype):
    H, W, T = 23, 31, 130                                 # 130 chunks: more than the default node holds
    rng = np.random.default_rng(3)
    maps = rng.uniform(0, 500, (T, H, W))
    maps[:, rng.uniform(size=(H, W)) < 0.3] = np.nan
    x = np.arange(W) * 5000.0 + 2500.0
    y = (np.arange(H) * 5000.0 + 2500.0)[::-1]
    path = str(tmp_path / "dis.nc")
    O.write_netcdf4(path, "dis", maps, x, y, time_values=np.arange(T) * 1.0, dtype=dtype, standard_name="DischargeMaps",
                    long_name="ChanQAvg", units="m3/s", projection=("laea", {"grid_mapping_name": "lambert_azimuthal_equal_area"}),
                    coord_attrs={"x": {"units": "Meter"}}, esri_pe_string="PROJCS[...]")
    got, gx, gy, gt = O.read_netcdf4(path, "dis")
    want = maps.astype(dtype).astype(np.float64)
    assert np.array_equal(np.isnan(got), np.isnan(maps)) and np.array_equal(got[~np.isnan(got)], want[~np.isnan(maps)])
    assert np.array_equal(gx, x) and np.array_equal(gy, y) and np.array_equal(gt, np.arange(T) * 1.0)
    r = H5.read(path)
    a = r.attrs("dis")
    assert a["units"] == "m3/s" and a["long_name"] == "ChanQAvg" and list(a["_Netcdf4Coordinates"]) == [2, 1, 0]
    assert [int(r.attrs(n)["_Netcdf4Dimid"]) for n in ("x", "y", "time")] == [0, 1, 2]
    assert r.attrs("x")["units"] == "Meter" and r.attrs()["Conventions"] == "CF-1.6"
    raw = open(path, "rb").read()
    assert raw[:8] == H5.SIGNATURE and raw[8] == 1
    assert struct.unpack_from("<Q", raw, 44)[0] == len(raw)               # end-of-file address of the superblock
    # the references: REFERENCE_LIST of a scale points at the variable's object header, DIMENSION_LIST at the scales
    kind, cls, shape, body = r.attrs("time")["REFERENCE_LIST"]
    assert cls == 6 and shape == (1,) and struct.unpack_from("<Qi", body)[0] == r.objects["dis"] and struct.unpack_from("<Qi", body)[1] == 0
    kind, cls, shape, body = a["DIMENSION_LIST"]
    assert cls == 9 and shape == (3,)
    gcol = struct.unpack_from("<IQI", body)[1]
    assert raw[gcol:gcol + 4] == b"GCOL"
    refs = [struct.unpack_from("<Q", raw, gcol + 16 + 24 * i + 16)[0] for i in range(3)]
    assert refs == [r.objects["time"], r.objects["y"], r.objects["x"]]


def test_single_map_and_state_maps(tmp_path):
    H, W = 12, 9
    rng = np.random.default_rng(8)
    mask = rng.uniform(size=(H, W)) < 0.7
    m = np.where(mask, rng.uniform(0, 3, (H, W)), np.nan)
    path = str(tmp_path / "end.nc")
    O.write_netcdf4(path, "ChanQEnd", m, np.arange(W, dtype=float), np.arange(H, dtype=float)[::-1])
    got = O.read_netcdf4(path, "ChanQEnd")
    assert got[3] is None and np.array_equal(np.isnan(got[0]), ~mask) and np.array_equal(got[0][mask], m[mask])
    state = {"ChanQState": rng.uniform(0, 5, int(mask.sum())), "LakeLevelState": np.full(int(mask.sum()), -9999.0)}
    for fmt in ("netcdf4", "classic"):
        d = str(tmp_path / fmt)
        O.write_state_maps(d, state, mask, fmt=fmt)
        back = O.read_state_maps(d, mask)
        assert sorted(back) == sorted(state)
        for k in state:
            assert np.array_equal(back[k], state[k]), (fmt, k)
    with open(os.path.join(str(tmp_path / "netcdf4"), "ChanQState.nc"), "rb") as f:
        assert f.read(8) == H5.SIGNATURE


def test_same_structure_as_the_reference_dis_nc(tmp_path):
    """tests/golden/ref_disnc_structure.json: the structure of the dis.nc the reference itself wrote for LF_ETRS89_UseCase
    (reference/output_reference_daily/dis.nc, extracted with h5py by make_hdf5_golden.py refstruct).  write_netcdf4 given
    the same shapes and metadata must produce the same variables, dimensions, types, chunking, filters, fill value and
    attributes (the global provenance attributes differ by design: no institution / creator_name of the JRC)."""
    import json
    ref = json.load(open(os.path.join(os.path.dirname(GOLD), "ref_disnc_structure.json")))
    v = ref["variables"]
    T, H, W = v["dis"]["shape"]
    rng = np.random.default_rng(4)
    maps = rng.uniform(0, 80, (T, H, W))
    maps[:, rng.uniform(size=(H, W)) < 0.45] = np.nan
    path = str(tmp_path / "dis.nc")
    da = v["dis"]["attrs"]
    O.write_netcdf4(path, "dis", maps, np.arange(W) * 5000.0, np.arange(H)[::-1] * 5000.0, time_values=np.arange(T) * 1.0,
                    time_units=v["time"]["attrs"]["units"], calendar=v["time"]["attrs"]["calendar"],
                    standard_name=da["standard_name"], long_name=da["long_name"], units=da["units"],
                    esri_pe_string=da["esri_pe_string"], coord_attrs={"x": v["x"]["attrs"], "y": v["y"]["attrs"]})
    r = H5.read(path)
    assert sorted(r.objects) == sorted(v)
    for name, want in v.items():
        got = r.info(name)
        assert list(got["shape"]) == want["shape"] and np.dtype(got["dtype"]) == np.dtype(want["dtype"]), name
        assert (list(got["chunks"]) if got["chunks"] else None) == want["chunks"], name
        filt = dict(got["filters"])
        assert (1 in filt) == (want["compression"] == "gzip") and (2 in filt) == want["shuffle"], name
        if want["compression"] == "gzip":
            assert filt[1] == [want["compression_opts"]] and filt[2] == [8]
        a = r.attrs(name)
        for k, val in want["attrs"].items():
            assert k in a, (name, k)
            if isinstance(val, str):
                assert a[k] == val, (name, k)
            else:
                assert np.allclose(np.asarray(a[k], float).reshape(-1), np.asarray(val, float).reshape(-1)), (name, k)
    assert float(r.info("dis")["fill"]) == v["dis"]["fillvalue"] == -9999.0
    # dimensions of the value variable, in the reference's order, through the netCDF-4 dimension ids
    ids = {int(r.attrs(n)["_Netcdf4Dimid"]): n for n in ("x", "y", "time")}
    assert [ids[i] for i in r.attrs("dis")["_Netcdf4Coordinates"]] == v["dis"]["dims"]
    assert set(ref["root_attrs"]) - {"institution", "creator_name"} <= set(r.attrs())


def test_chunks_libhdf5_never_allocated_read_as_the_fill_value():
    """a file written by libhdf5 (tests/golden/make_hdf5_golden.py sparse) in which two of four chunks of a variable and
    all chunks of another were never written: they read as the fill value (-9999 = LISFLOOD's cold-start marker), not 0"""
    r = H5.read(os.path.join(os.path.dirname(GOLD), "h5py_sparse.h5"))
    a = r.dataset("state")
    base = np.arange(15, dtype="f8").reshape(3, 5)
    assert np.array_equal(a[0], base) and np.array_equal(a[2], base * 2.0)
    assert (a[1] == -9999.0).all() and (a[3] == -9999.0).all()
    n = r.dataset("never")
    assert n.shape == (2, 3) and n.dtype == np.float32 and (n == np.float32(-1.5)).all()


def test_streamed_map_stack_equals_the_one_written_at_once(tmp_path, monkeypatch):
    """output.NetCDF4MapWriter appends one step's chunk at a time (the stack is never in memory): with every step written
    in order the file is byte for byte the one write_netcdf4 makes from the whole stack"""
    import time
    monkeypatch.setattr(time, "ctime", lambda *a: "Thu Jan  1 00:00:00 1970")
    H, W, T = 19, 27, 40
    rng = np.random.default_rng(8)
    maps = rng.uniform(0, 500, (T, H, W))
    maps[:, rng.uniform(size=(H, W)) < 0.3] = np.nan
    x, y, tv = np.arange(W) * 1.0, np.arange(H)[::-1] * 1.0, np.arange(T) * 1.0
    kw = dict(dtype="f4", units="m3/s", projection=("laea", {"grid_mapping_name": "lambert_azimuthal_equal_area"}))
    O.write_netcdf4(str(tmp_path / "a.nc"), "dis", maps, x, y, time_values=tv, **kw)
    with O.NetCDF4MapWriter(str(tmp_path / "b.nc"), "dis", x, y, time_values=tv, **kw) as w:
        for t in range(T):
            w.write_step(t, maps[t])
    assert open(tmp_path / "a.nc", "rb").read() == open(tmp_path / "b.nc", "rb").read()
    # a single map (no time axis)
    O.write_netcdf4(str(tmp_path / "c.nc"), "dis", maps[0], x, y, **kw)
    with O.NetCDF4MapWriter(str(tmp_path / "d.nc"), "dis", x, y, **kw) as w:
        w.write_step(0, maps[0])
    assert open(tmp_path / "c.nc", "rb").read() == open(tmp_path / "d.nc", "rb").read()


def test_streamed_map_stack_out_of_order_flush_and_missing_steps(tmp_path):
    H, W, T = 11, 13, 9
    rng = np.random.default_rng(2)
    maps = rng.uniform(0, 5, (T, H, W))
    x, y = np.arange(W) * 1.0, np.arange(H)[::-1] * 1.0
    path = str(tmp_path / "s.nc")
    w = O.NetCDF4MapWriter(path, "dis", x, y, time_values=np.arange(T) * 1.0)
    for t in (4, 0, 8):
        w.write_step(t, maps[t])
    w.flush()                                                              # readable while still open
    got = O.read_netcdf4(path, "dis")[0]
    assert np.array_equal(got[[4, 0, 8]], maps[[4, 0, 8]]) and np.isnan(got[[1, 2, 3, 5, 6, 7]]).all()
    raw = open(path, "rb").read()
    assert struct.unpack_from("<Q", raw, 44)[0] == len(raw)
    for t in (2, 1, 7):
        w.write_step(t, maps[t])
    with pytest.raises(ValueError):
        w.write_step(2, maps[2])                                           # a chunk is written once
    with pytest.raises(IndexError):
        w.write_step(T, maps[0])
    with pytest.raises(ValueError):
        w.write_step(3, maps[3][:, :5])
    w.close()
    w.close()                                                              # idempotent
    got = O.read_netcdf4(path, "dis")[0]
    done = [0, 1, 2, 4, 7, 8]
    assert np.array_equal(got[done], maps[done]) and np.isnan(got[[3, 5, 6]]).all()
    with pytest.raises(ValueError):
        H5.write(str(tmp_path / "x.h5"), [H5.Dataset("v", None, chunks=(1, 2), shape=(3, 2), dtype="f8")])
    # a streamed dataset whose edge chunks are partial blocks
    d = H5.Dataset("v", None, chunks=(4, 4), shape=(6, 7), dtype="f4", fill=np.float32(-1))
    a = rng.uniform(0, 1, (6, 7)).astype("f4")
    with H5.Writer(str(tmp_path / "e.h5"), [d]) as hw:
        for i in range(2):
            for j in range(2):
                hw.write_chunk("v", (i, j), a[4 * i:4 * i + 4, 4 * j:4 * j + 4])
    assert np.array_equal(H5.read(str(tmp_path / "e.h5")).dataset("v"), a)


def test_two_streamed_datasets_in_one_file(tmp_path):
    """chunks of several streamed datasets interleave behind the metadata; each keeps its own chunk index"""
    rng = np.random.default_rng(5)
    a = rng.uniform(0, 1, (5, 6, 7))
    b = rng.integers(0, 100, (3, 4)).astype(np.int32)
    ds = [H5.Dataset("t", np.arange(5.0), ("t",)),
          H5.Dataset("a", None, chunks=(1, 6, 7), deflate=4, shuffle=True, fill=np.float64(-1), shape=a.shape, dtype="f8"),
          H5.Dataset("b", None, chunks=(1, 4), shape=b.shape, dtype="i4"),
          H5.Dataset("c", np.arange(4, dtype=np.float32))]
    path = str(tmp_path / "two.h5")
    with H5.Writer(path, ds) as w:
        for i in range(5):
            w.write_chunk("a", (i, 0, 0), a[i][None])
            if i < 3:
                w.write_chunk("b", (i, 0), b[i][None])
    r = H5.read(path)
    assert np.array_equal(r.dataset("a"), a) and np.array_equal(r.dataset("b"), b)
    assert np.array_equal(r.dataset("c"), np.arange(4, dtype=np.float32)) and np.array_equal(r.dataset("t"), np.arange(5.0))
    with pytest.raises(KeyError):
        H5.Writer(str(tmp_path / "x.h5"), ds[:1]).write_chunk("a", (0, 0, 0), a[0][None])


def test_shape_error_leaves_no_file_and_a_dropped_writer_keeps_its_steps(tmp_path):
    """write_netcdf4 checks the map shape before it creates (or truncates) the file; a NetCDF4MapWriter that is dropped
    without close() -- an exception in the caller's step loop -- still completes the chunk index on garbage collection"""
    import gc
    from lisflood_amd import output
    x, y = np.arange(5.0), np.arange(4.0)
    path = str(tmp_path / "bad.nc")
    with pytest.raises(ValueError):
        output.write_netcdf4(path, "dis", np.zeros((4, 6)), x, y)
    assert not os.path.exists(path)
    path = str(tmp_path / "dropped.nc")
    maps = np.random.default_rng(3).random((3, 4, 5))
    w = output.NetCDF4MapWriter(path, "dis", x, y, time_values=[0.0, 1.0, 2.0])
    w.write_step(0, maps[0])
    w.write_step(1, maps[1])
    del w
    gc.collect()
    got = output.read_netcdf4(path, "dis")[0]
    np.testing.assert_array_equal(got[:2], maps[:2])
    assert np.isnan(got[2]).all()
