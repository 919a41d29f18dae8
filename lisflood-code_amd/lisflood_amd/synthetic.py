"""Seeded synthetic catchments for parity tests and bench.py.

LDD families (all acyclic by construction -- every cell drains to a strictly lower neighbour of a
potential phi, or is a pit):
  shallow : phi = U(0,1)                                   -> ~11 % pits, NL ~ 7-8 ("random LDD")
  deep    : phi = row-from-bottom + 0.25*col/W + 0.6*U(0,1) -> NL = H+2, ~W cells per level (sheet flow)
  saddle  : like deep, but the right half of the raster drains upward (tests: flow crosses a row cut both ways)
  river   : phi = distance to the nearest of M seeded outlets + 0.75*U(0,1) (M ~ one per 2.5 Mcells): convergent,
            dendritic drainage towards a few outlets -- NL ~ basin radius (hundreds to thousands of levels), level
            widths from 1 cell next to an outlet to tens of thousands: the regime of a real river network
Codes follow the LISFLOOD/PCRaster keypad convention of the reference
(kinematic_wave_parallel.py:49-51): index 0..7 <-> codes [2,3,6,9,8,7,4,1], 5 = pit, and the
(row, col) shifts IX_ADDS.  Ties are resolved to the first minimum in IX_ADDS order.

The noise of row r depends only on (seed, r // ROWS_PER_CHUNK), so a rank of a row-block partition can
generate exactly its own rows (plus halo rows) of the global raster without generating the rest.
"""
import numpy as np

IX_ADDS = ((1, 0), (1, 1), (0, 1), (-1, 1), (-1, 0), (-1, -1), (0, -1), (1, -1))
FLOW_CODE = (2, 3, 6, 9, 8, 7, 4, 1, 5)
ROWS_PER_CHUNK = 256
FAMILIES = ("shallow", "deep", "river")


def river_outlets(H, W, seed):
    """(rows, cols) of the outlets of the `river` family: M = max(3, H*W / 2.5e6) seeded positions"""
    m = max(3, int(round(H * W / 2.5e6)))
    rng = np.random.default_rng([int(seed), 777])
    return rng.integers(0, H, m).astype(np.float64), rng.integers(0, W, m).astype(np.float64)


def _noise_rows(seed, r0, r1, W):
    """U(0,1) noise for global rows [r0, r1) -- chunk-seeded so any row range is reproducible."""
    out = np.empty((r1 - r0, W))
    c0 = r0 // ROWS_PER_CHUNK
    c1 = (r1 - 1) // ROWS_PER_CHUNK
    for c in range(c0, c1 + 1):
        rng = np.random.default_rng([int(seed), int(c)])
        blk = rng.random((ROWS_PER_CHUNK, W))
        a = max(r0, c * ROWS_PER_CHUNK)
        b = min(r1, (c + 1) * ROWS_PER_CHUNK)
        out[a - r0:b - r0] = blk[a - c * ROWS_PER_CHUNK:b - c * ROWS_PER_CHUNK]
    return out


def potential_rows(family, H, W, seed, r0, r1):
    """phi for global rows [r0, r1) of an H x W raster."""
    u = _noise_rows(seed, r0, r1, W)
    if family == "shallow":
        return u
    if family == "river":
        orow, ocol = river_outlets(H, W, seed)
        rr = np.arange(r0, r1, dtype=np.float64)[:, None]
        cc = np.arange(W, dtype=np.float64)[None, :]
        d = np.full((r1 - r0, W), np.inf)
        for a, b in zip(orow, ocol):
            np.minimum(d, np.sqrt((rr - a) ** 2 + (cc - b) ** 2), out=d)
        return d + 0.75 * u
    rows_from_bottom = (H - 1 - np.arange(r0, r1, dtype=np.float64))[:, None]
    cols = np.arange(W, dtype=np.float64)[None, :]
    if family == "deep":
        return rows_from_bottom + 0.25 * cols / W + 0.6 * u
    if family == "saddle":   # left half drains to the bottom row, right half to the top row
        tilt = np.where(cols < W / 2, rows_from_bottom, (H - 1) - rows_from_bottom)
        return tilt + 0.25 * cols / W + 0.6 * u
    raise ValueError("unknown LDD family %r" % (family,))


def make_ldd(family, H, W, seed, r0=0, r1=None, land_mask=None):
    """uint8 LDD codes for global rows [r0, r1) of the H x W synthetic raster (all land unless
    land_mask[H, W] is given: non-land cells get code 0 and never receive flow)."""
    if r1 is None:
        r1 = H
    out = np.empty((r1 - r0, W), np.uint8)
    step = 1024
    for a in range(r0, r1, step):
        b = min(r1, a + step)
        ga, gb = max(0, a - 1), min(H, b + 1)
        phi = potential_rows(family, H, W, seed, ga, gb)
        if land_mask is not None:
            phi = np.where(land_mask[ga:gb], phi, np.inf)
        pad = np.full((gb - ga + 2, W + 2), np.inf)
        pad[1:-1, 1:-1] = phi
        # rows of `pad` that correspond to global rows [a, b)
        o = a - ga + 1
        n = b - a
        centre = pad[o:o + n, 1:-1]
        best = centre.copy()
        code = np.full((n, W), FLOW_CODE[8], np.uint8)
        for k, (dr, dc) in enumerate(IX_ADDS):
            nb = pad[o + dr:o + dr + n, 1 + dc:1 + dc + W]
            better = nb < best          # strict: first minimum in IX_ADDS order wins ties
            best = np.where(better, nb, best)
            code[better] = FLOW_CODE[k]
        if land_mask is not None:
            code[~land_mask[a:b]] = 0
        out[a - r0:b - r0] = code
    return out


def router_params(N, seed=3, beta=0.6, dt=3600.0):
    """alpha ~ logU(0.4,16), dx ~ U(500,15000), Q0 = (A0/alpha)^(1/beta) with A0 ~ logU(0.05,500)."""
    rng = np.random.default_rng(seed)
    alpha = np.exp(rng.uniform(np.log(0.4), np.log(16.0), N))
    dx = rng.uniform(500.0, 15000.0, N)
    a0 = np.exp(rng.uniform(np.log(0.05), np.log(500.0), N))
    q0 = (a0 / alpha) ** (1.0 / beta)
    return dict(alpha=alpha, dx=dx, Q0=q0, beta=beta, dt=dt)


def lateral_inflow(N, step, seed=4, hi=2e-4):
    """specific lateral inflow q ~ U(0, 2e-4) [m3 s-1 m-1], redrawn each step (seed 4 + step)."""
    return np.random.default_rng(seed + step).uniform(0.0, hi, N)


def model_step_values_slice(N, i0, i1, p_slice, seed=17):
    """Elements [i0, i1) of model_step_values(N, router_params(N), seed)[0] without generating the rest
    (`p_slice` = router_params_slice(N, i0, i1)): what one rank of a row-block partition needs."""
    n = i1 - i0
    beta, dt = p_slice["beta"], 3600.0
    alpha, length, q0 = p_slice["alpha"], p_slice["dx"], p_slice["Q0"]
    alpha2 = alpha * (1.2 + (2.0 - 1.2) * _stream_slice(seed, i0, n))
    qlimit = 2.0 * q0 * (0.3 + (1.2 - 0.3) * _stream_slice(seed, N + i0, n))
    vals = dict(ChanLength=length, InvChanLength=1 / length, ChannelAlpha=alpha, InvChannelAlpha=1 / alpha,
                ChannelAlpha2=alpha2, InvChannelAlpha2=1 / alpha2, QLimit=qlimit, M3Limit=alpha * length * qlimit ** beta,
                Chan2M3Start=alpha2 * length * qlimit ** beta, Chan2QStart=qlimit * 0.1, PixelArea=np.full(n, 2.5e7),
                IsChannelKinematic=np.ones(n, bool), SideflowChanM3=lateral_inflow_slice(N, 0, i0, i1) * length * dt)
    vals["Chan2M3Kin"] = vals["Chan2M3Start"].copy()
    vals["ChanM3Kin"] = alpha * length * q0 ** beta
    vals["ChanQKin"] = q0.copy()
    vals["Chan2QKin"] = (vals["Chan2M3Kin"] / length / alpha2) ** (1 / beta)
    return vals, dt


def _stream_slice(seed, offset, count):
    """`count` uniform(0,1) doubles starting at element `offset` of default_rng(seed)'s stream (PCG64 jump-ahead:
    one 64-bit draw per double), so a rank can draw exactly its slice of a global vector."""
    bg = np.random.PCG64(seed)
    bg.advance(int(offset))
    return np.random.Generator(bg).random(int(count))


def router_params_slice(N, i0, i1, seed=3, beta=0.6, dt=3600.0):
    """Elements [i0, i1) of router_params(N, seed) without generating the rest."""
    n = i1 - i0
    u1, u2, u3 = (_stream_slice(seed, k * N + i0, n) for k in range(3))
    la, lb = np.log(0.4), np.log(16.0)
    alpha = np.exp(la + (lb - la) * u1)
    dx = 500.0 + (15000.0 - 500.0) * u2
    la0, lb0 = np.log(0.05), np.log(500.0)
    a0 = np.exp(la0 + (lb0 - la0) * u3)
    q0 = (a0 / alpha) ** (1.0 / beta)
    return dict(alpha=alpha, dx=dx, Q0=q0, beta=beta, dt=dt)


def lateral_inflow_slice(N, step, i0, i1, seed=4, hi=2e-4):
    return 0.0 + (hi - 0.0) * _stream_slice(seed + step, i0, i1 - i0)


def soil_params(N, V=3, L=3, seed=11, frozen_frac=0.1, zero_pore_frac=0.02):
    """Synthetic soil-column inputs in the layout of soilColumnsWaterBalance (soilloop.py:79-99).

    Returns a dict whose keys are exactly the reference's argument names.  Ranges follow SURVEY.md
    section 8(d): theta_s 0.35-0.5, theta_r 0.02-0.08, lambda 0.1-0.4, KSat 5-500 mm/d, depths
    50 / 100-600 / 300-1500 mm, b 0.1-0.8, Rain 0-30 mm, 10 % frozen.
    """
    rng = np.random.default_rng(seed)
    d = {}
    depth1a = np.full((L, N), 50.0)
    depth1b = rng.uniform(100.0, 600.0, (L, N))
    depth2 = rng.uniform(300.0, 1500.0, (L, N))
    zero = rng.random((L, N)) < zero_pore_frac
    depth1b = np.where(zero, 0.0, depth1b)
    for name, depth in (("1a", depth1a), ("1b", depth1b), ("2", depth2)):
        ths = rng.uniform(0.35, 0.5, (L, N))
        thr = rng.uniform(0.02, 0.08, (L, N))
        lam = rng.uniform(0.1, 0.4, (L, N))
        m = lam / (lam + 1.0)
        d["SoilDepth" + name] = depth
        d["WS" + name] = ths * depth
        d["WRes" + name] = thr * depth
        fc = thr + (ths - thr) * rng.uniform(0.45, 0.7, (L, N))
        wp = thr + (ths - thr) * rng.uniform(0.1, 0.3, (L, N))
        d["WFC" + name] = fc * depth
        d["WWP" + name] = wp * depth
        d["KSat" + name] = np.exp(rng.uniform(np.log(5.0), np.log(500.0), (L, N)))
        d["GenuM" + name] = m
        d["GenuInvM" + name] = 1.0 / m
        d["PoreSpaceNotZero" + name] = (ths * depth) != 0
    for k in ("WS", "WRes", "WFC", "WWP"):
        d[k + "1"] = d[k + "1a"] + d[k + "1b"]
    d["StoreMaxPervious"] = d["WS1"] / rng.uniform(0.2, 1.0, (L, N)) * 0.5
    sat = rng.uniform(0.05, 1.0, (3, V, N))
    lu = np.arange(V) % L
    d["index_landuse_all"] = lu.astype(np.int64)
    d["is_irrigated"] = np.array([(v % L) == 2 for v in range(V)], bool)
    d["is_paddy_irrig"] = np.zeros(V, bool)
    d["paddy_inactive"] = np.zeros((1, N), bool)
    for i, name in enumerate(("1a", "1b", "2")):
        d["W" + name] = d["WRes" + name][lu] + sat[i] * (d["WS" + name][lu] - d["WRes" + name][lu])
    d["W1"] = d["W1a"] + d["W1b"]
    d["DSLR"] = 1.0 + np.floor(rng.uniform(0.0, 6.0, (V, N)))
    d["UZ"] = rng.uniform(0.0, 20.0, (V, N))
    d["LeafDrainage"] = rng.uniform(0.0, 2.0, (V, N))
    d["Interception"] = rng.uniform(0.0, 1.0, (V, N))
    d["ESMax"] = rng.uniform(0.0, 4.0, (V, N))
    rain = rng.uniform(0.0, 30.0, N)
    rain[rng.random(N) < 0.4] = 0.0
    d["Rain"] = rain
    d["SnowMelt"] = np.where(rng.random(N) < 0.1, rng.uniform(0.0, 5.0, N), 0.0)
    d["isFrozenSoil"] = rng.random(N) < frozen_frac
    d["b_Xinanjiang"] = rng.uniform(0.1, 0.8, N)
    d["PowerInfPot"] = rng.uniform(1.0, 3.0, N)
    d["PowerPrefFlow"] = rng.uniform(1.0, 6.0, N)
    d["UpperZoneK"] = rng.uniform(0.01, 0.3, N)
    d["GwPercStep"] = rng.uniform(0.1, 1.5, N)
    d["DtDay"] = 1.0
    d["AvWaterThreshold"] = 1.0
    d["CourantCrit"] = 0.4
    d["DrainedFraction"] = 0.1
    for k in ("AvailableWaterForInfiltration", "ESAct", "PrefFlow", "Infiltration", "Theta1a", "Theta1b",
              "Theta2", "Sat1a", "Sat1b", "Sat1", "Sat2", "SeepTopToSubA", "SeepTopToSubB", "SeepSubToGW",
              "UZOutflow", "GwPercUZLZ"):
        d[k] = np.zeros((V, N))
    return d


# positional order of soilColumnsWaterBalance (soilloop.py:79-99)
SOIL_ARG_ORDER = (
    "index_landuse_all is_irrigated is_paddy_irrig paddy_inactive DtDay AvailableWaterForInfiltration Rain "
    "SnowMelt LeafDrainage Interception DSLR AvWaterThreshold ESAct ESMax isFrozenSoil b_Xinanjiang "
    "StoreMaxPervious PowerInfPot PrefFlow PowerPrefFlow Infiltration CourantCrit PoreSpaceNotZero1a "
    "PoreSpaceNotZero1b PoreSpaceNotZero2 KSat1a KSat1b KSat2 GenuInvM1a GenuInvM1b GenuInvM2 GenuM1a GenuM1b "
    "GenuM2 W1a W1b W1 W2 Theta1a Theta1b Theta2 Sat1a Sat1b Sat1 Sat2 SeepTopToSubA SeepTopToSubB SeepSubToGW "
    "WRes1a WRes1b WRes1 WRes2 WWP1a WWP1b WWP1 WWP2 WFC1a WFC1b WFC1 WFC2 SoilDepth1a SoilDepth1b SoilDepth2 "
    "WS1a WS1b WS1 WS2 UpperZoneK DrainedFraction GwPercStep UZOutflow UZ GwPercUZLZ").split()

SOIL_WRITTEN = (
    "AvailableWaterForInfiltration DSLR ESAct PrefFlow Infiltration W1a W1b W1 W2 Theta1a Theta1b Theta2 Sat1a "
    "Sat1b Sat1 Sat2 SeepTopToSubA SeepTopToSubB SeepSubToGW UZOutflow UZ GwPercUZLZ").split()


def interception_params(N, V=3, seed=21):
    rng = np.random.default_rng(seed)
    lai = rng.uniform(0.0, 8.0, (V, N))
    lai[:, : N // 20] = rng.uniform(0.0, 0.1, (V, N // 20))          # SMax = 0 branch
    lai[:, N // 20: N // 10] = rng.uniform(43.3, 50.0, (V, N // 10 - N // 20))  # saturated branch
    rain = rng.uniform(0.0, 30.0, N)
    rain[rng.random(N) < 0.4] = 0.0
    return dict(
        Interception=np.zeros((V, N)), TaInterception=np.zeros((V, N)), LeafDrainage=np.zeros((V, N)),
        CumInterception=rng.uniform(0.0, 3.0, (V, N)) * (rng.random((V, N)) < 0.7),
        LAI=lai, Rain=rain, TaInterceptionMax=rng.uniform(0.0, 3.0, (V, N)), drainageK=0.25)


def hotpath_scenario(H, W, seed=101, channel_frac=0.3, nsteps=24, dt_sec=86400.0, family="deep", block=None):
    """A complete synthetic input set for the device-resident hot path (lisflood_amd.hotpath.HotPathDevice) and
    for the module classes: -> (values, scalars, land_mask, ldd_to_chan, ldd_kinematic).  All-land H x W raster of the
    LDD `family`, `channel_frac` of the cells are channel pixels (so overland routing between cells is exercised).
    block: draw the per-pixel parameter and state fields for `block` pixels only and repeat them over the raster (the LDD,
    the channel mask and everything derived from them are always drawn at full size) -- at 5000^2 the ~110 fields are 48 GB
    and drawing them takes minutes; repeated fields stream through the kernels exactly like fresh ones."""
    from . import ldd as L
    rng = np.random.default_rng(seed)
    N_full = H * W
    mask = np.ones((H, W), bool)
    codes = make_ldd(family, H, W, seed)[mask].astype(np.float64)
    is_chan_full = rng.random(N_full) < channel_frac
    ldd_to_chan = np.where(is_chan_full, 5.0, codes)
    kin_codes, _ = L.lddmask(codes, mask, is_chan_full)
    ldd_kin = np.zeros(N_full)
    ldd_kin[is_chan_full] = kin_codes
    del codes, kin_codes
    N = N_full if not block or block >= N_full else int(block)      # the fields below are drawn for N pixels
    is_chan = is_chan_full[:N]
    beta = 0.6
    dt_routing = dt_sec / nsteps
    v = {}
    soil = soil_params(N, seed=seed + 1)
    for k, a in soil.items():
        if isinstance(a, np.ndarray) and a.ndim >= 1 and k not in ("index_landuse_all", "is_irrigated", "is_paddy_irrig",
                                                                      "paddy_inactive"):
            v[k] = a
    ip = interception_params(N, seed=seed + 2)
    v["LAI"] = ip["LAI"]
    v["CumInterception"] = ip["CumInterception"]
    v["TaInterception"] = np.zeros((3, N))
    v["LAITerm"] = np.exp(-0.5 * ip["LAI"])
    v["CropCoef"] = rng.uniform(0.6, 1.2, (3, N))
    v["CropGroupNumber"] = np.stack([rng.uniform(1, 5, N), rng.uniform(1, 5, N), np.full(N, 2.0)])
    for k in ("potential_transpiration", "RWS", "Ta"):
        v[k] = np.zeros((3, N))
    frac = rng.dirichlet([3, 2, 1], N).T * rng.uniform(0.5, 1.0, N)
    v["SoilFraction"] = frac
    v["SoilDepthTotal"] = soil["SoilDepth1a"] + soil["SoilDepth1b"] + soil["SoilDepth2"]
    v["DirectRunoffFraction"] = rng.uniform(0, 0.15, N) * (rng.random(N) < 0.5)
    v["WaterFraction"] = rng.uniform(0, 0.1, N) * (rng.random(N) < 0.3)
    v["SMaxSealed"] = np.full(N, 1.0)
    v["LowerZoneK"] = rng.uniform(0.001, 0.05, N)
    v["LZThreshold"] = np.zeros(N)
    v["GwLossStep"] = np.zeros(N)
    v["LZ"] = rng.uniform(0, 100, N)
    for k in ("CumInterSealed", "LZInflowCUM", "TaInterceptionCUM", "TaCUM", "ESActCUM", "GwLossCUM"):
        v[k] = np.zeros(N)
    pixel_length, pixel_area = 5000.0, 2.5e7
    grad = rng.uniform(0.001, 0.2, N)
    nman = np.stack([rng.uniform(0.05, 0.2, N), rng.uniform(0.2, 0.5, N), rng.uniform(0.01, 0.05, N)])
    v["OFAlpha"] = ((nman / np.sqrt(grad)) ** beta) * ((pixel_length + 2 * 0.001 * 5.0) ** (2.0 / 3.0 * beta))
    for k in ("OFQDirect", "OFQOther", "OFQForest"):
        v[k] = rng.uniform(0, 0.3, N)
    # ---- from here on at full size: the block's draws repeated over the raster, the channel mask the full-size one ----
    reps = -(-N_full // N)
    rep = (lambda a: a) if N == N_full else (lambda a: np.ascontiguousarray(np.tile(a, reps)[..., :N_full]))
    for k in list(v):
        v[k] = rep(v[k])
    p = {k: (rep(a) if isinstance(a, np.ndarray) else a) for k, a in router_params(N, seed=seed + 3).items()}
    u_alpha2 = rep(rng.uniform(1.2, 2.0, N))
    Nb, is_chan, N = N, is_chan_full, N_full
    v["IsChannel"] = is_chan
    alpha = np.where(is_chan, p["alpha"], 1.0)
    length = p["dx"]
    alpha2 = alpha * u_alpha2
    q0 = np.where(is_chan, np.minimum(p["Q0"], 500.0), 0.0)
    qlimit = 2.0 * q0 * rep(rng.uniform(0.3, 1.2, Nb))
    v.update(ChannelAlpha=alpha, InvChannelAlpha=1 / alpha, ChannelAlpha2=alpha2, InvChannelAlpha2=1 / alpha2,
             ChanLength=length, InvChanLength=1 / length, QLimit=qlimit, M3Limit=alpha * length * qlimit ** beta,
             Chan2M3Start=alpha2 * length * qlimit ** beta, Chan2QStart=0.1 * qlimit, PixelArea=np.full(N, pixel_area),
             IsChannelKinematic=is_chan.copy())
    v["Chan2M3Kin"] = v["Chan2M3Start"].copy()
    v["ChanM3Kin"] = alpha * length * q0 ** beta
    v["ChanQKin"] = q0.copy()
    v["Chan2QKin"] = (v["Chan2M3Kin"] / length / alpha2) ** (1 / beta)
    v["ChanQ"] = np.maximum(v["ChanQKin"] + v["Chan2QKin"] - qlimit, 0.0)
    for k in ("CrossSection2Area", "Sideflow1Chan", "sumDisDay"):
        v[k] = np.zeros(N)
    scalars = dict(Beta=beta, DtSec=dt_sec, DtRouting=dt_routing, NoRoutSteps=nsteps, DtDay=1.0, InvDtDay=1.0,
                   PixelLength=pixel_length, MMtoM3=0.001 * pixel_area, M3toMM=1 / (0.001 * pixel_area),
                   LeafDrainageK=0.25, AvWaterThreshold=1.0, CourantCrit=0.4, DrainedFraction=0.1)
    return v, scalars, mask, ldd_to_chan, ldd_kin


def hotpath_forcing(N, step, seed=300):
    rng = np.random.default_rng(seed + step)
    return dict(Rain=rng.uniform(0, 20, N) * (rng.random(N) < 0.6), SnowMelt=np.zeros(N),
                EWRef=rng.uniform(0, 5, N), ETRef=rng.uniform(0, 4, N), ESRef=rng.uniform(0, 3, N))


def structures_scenario(codes, shape, chan_q, dt_routing, n_lakes=64, n_res=192, seed=23):
    """Synthetic lakes / reservoirs / inflow points / transmission-loss reaches on an all-land compressed LDD
    (parameter ranges as tests/golden/make_golden.py used on LF_ETRS89: lakes.py:96-160, reservoir.py:73-165).
    Returns (var attributes dict, cut LDD codes): the cells just upstream of a site are pits of the cut LDD
    (structures.py:44-61), `downstruct` keeps the uncut links (routing.py:159-164)."""
    from . import ldd as L
    H, W = shape
    N = H * W
    rng = np.random.default_rng(seed)
    mask = np.ones((H, W), bool)
    codes = np.asarray(codes, np.float64)
    down = L.downstream_index(codes, mask)
    nups = np.bincount(down[down >= 0], minlength=N)
    cand = np.nonzero((nups > 0) & (down >= 0))[0]                 # sites with something upstream and downstream
    sites = rng.choice(cand, n_lakes + n_res, replace=False)
    lake, res = np.sort(sites[:n_lakes]), np.sort(sites[n_lakes:])
    is_site = np.zeros(N, bool); is_site[sites] = True
    ups = (down >= 0) & is_site[np.maximum(down, 0)]
    cut = codes.copy(); cut[ups] = 5.0
    d = {}
    ds = np.where(down >= 0, down, N).astype(np.int32); ds[codes == 5] = N
    d["downstruct"] = ds
    d["LakeIndex"], d["ReservoirIndex"] = lake, res
    d["LakeAreaCC"] = rng.uniform(2e6, 5e7, n_lakes)
    lake_a = rng.uniform(5.0, 80.0, n_lakes)
    d["LakeFactor"] = d["LakeAreaCC"] / (dt_routing * np.sqrt(lake_a))
    d["LakeFactorSqr"] = np.square(d["LakeFactor"])
    qin = np.bincount(ds, weights=chan_q, minlength=N + 1)
    d["LakeInflowOldCC"] = qin[lake]
    d["LakeLevelCC"] = rng.uniform(0.5, 3.0, n_lakes)
    d["LakeStorageM3"] = np.zeros(N); d["LakeStorageM3"][lake] = d["LakeAreaCC"] * d["LakeLevelCC"]
    d["LakeOutflowCC"] = np.square(d["LakeLevelCC"]) * lake_a
    d["LakeStorageM3BalanceCC"] = d["LakeStorageM3"][lake].copy()
    d["TotalReservoirStorageM3CC"] = np.exp(rng.uniform(np.log(1e6), np.log(5e8), n_res))
    d["ConservativeStorageLimitCC"] = rng.uniform(0.05, 0.15, n_res)
    d["NormalStorageLimitCC"] = rng.uniform(0.4, 0.7, n_res)
    d["FloodStorageLimitCC"] = rng.uniform(0.8, 0.97, n_res)
    d["Normal_FloodStorageLimitCC"] = d["NormalStorageLimitCC"] + 0.5 * (d["FloodStorageLimitCC"] - d["NormalStorageLimitCC"])
    q0 = qin[res]
    d["MinReservoirOutflowCC"], d["NormalReservoirOutflowCC"] = 0.1 * q0 + 0.01, 0.9 * q0 + 0.05
    d["NonDamagingReservoirOutflowCC"] = 4.0 * q0 + 1.0
    d["DeltaO"] = d["NormalReservoirOutflowCC"] - d["MinReservoirOutflowCC"]
    d["DeltaLN"] = d["NormalStorageLimitCC"] - 2 * d["ConservativeStorageLimitCC"]
    d["DeltaNFL"] = d["FloodStorageLimitCC"] - d["Normal_FloodStorageLimitCC"]
    d["ReservoirStorageM3"] = np.zeros(N)
    d["ReservoirStorageM3"][res] = rng.uniform(0.02, 1.0, n_res) * d["TotalReservoirStorageM3CC"]
    d["QInM3Old"], d["QDelta"] = np.zeros(N), np.zeros(N)
    pts = rng.choice(N, 32, replace=False)
    d["QInM3Old"][pts] = rng.uniform(1e4, 2e5, 32); d["QDelta"][pts] = rng.uniform(0.0, 2e3, 32)
    d["UpTrans"] = (rng.random(N) < 0.3) & (chan_q > 1.0)
    # (Q^p2 - TransSub) must stay positive on every flagged reach for as long as the scenario is stepped
    d["TransPower1"], d["TransPower2"], d["TransSub"] = 1 / 0.95, 0.95, 1e-9
    d["TransCum"] = np.zeros(N)
    return d, cut


def model_step_values(N, p, seed=17, ids=None):
    """State and parameter vectors of a routing model step (split routing) for the synthetic benches and tests, as the
    dict `RoutingStepDevice` / `routing.var_from_values` take; `p` = router_params(N); `ids`: only these pixels."""
    rng = np.random.default_rng(seed)
    beta, dt = p["beta"], 3600.0
    # `ids`: every full-length vector is cut to the selection as soon as it is drawn (one or two of them alive at a time:
    # a rank of the catchment partition at 10 000^2 would otherwise hold ~17 vectors of 0.8 GB)
    cut = (lambda v: v) if ids is None else (lambda v: np.ascontiguousarray(v[ids]))
    n = N if ids is None else int(np.asarray(ids).size)
    alpha, length, q0 = cut(p["alpha"]), cut(p["dx"]), cut(p["Q0"])
    alpha2 = alpha * cut(rng.uniform(1.2, 2.0, N))
    qlimit = 2.0 * q0 * cut(rng.uniform(0.3, 1.2, N))
    vals = dict(ChanLength=length, InvChanLength=1 / length, ChannelAlpha=alpha, InvChannelAlpha=1 / alpha,
                ChannelAlpha2=alpha2, InvChannelAlpha2=1 / alpha2, QLimit=qlimit, M3Limit=alpha * length * qlimit ** beta,
                Chan2M3Start=alpha2 * length * qlimit ** beta, Chan2QStart=qlimit * 0.1, PixelArea=np.full(n, 2.5e7),
                IsChannelKinematic=np.ones(n, bool), SideflowChanM3=cut(lateral_inflow(N, 0)) * length * dt)
    vals["Chan2M3Kin"] = vals["Chan2M3Start"].copy()
    vals["ChanM3Kin"] = alpha * length * q0 ** beta
    vals["ChanQKin"] = q0.copy()
    vals["Chan2QKin"] = (vals["Chan2M3Kin"] / length / alpha2) ** (1 / beta)
    return vals, dt
