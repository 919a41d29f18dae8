"""Component layout on the GPU: the sweep by tiers / bins (one wavefront per bin) must give exactly the discharge of
the level sweep -- same per-cell arithmetic, same upstream summation order -- and both are pinned to the reference by
tests/test_gpu_parity.py."""
import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amd():
    import lisflood_amd
    from lisflood_amd import _lib
    if _lib.device_count() < 1:
        pytest.fail("no HIP device visible: the -m gpu tests need an MI355X")
    return lisflood_amd


@pytest.fixture(params=["fused_beta_3_5", "general_pow"])
def solver(request, monkeypatch):
    if request.param == "general_pow":
        monkeypatch.setenv("LF_GENERAL_POW", "1")
    return request.param


def test_components_route_etrs89_golden(amd, solver):
    """the reference's own vectors (route_etrs89.npz: 24 sub-steps, both sections) through the component layout"""
    from lisflood_amd.kinematic_wave_parallel import kinematicWave
    g = golden("route_etrs89")
    kw = kinematicWave(g["codes"], g["mask"], g["alpha"], float(g["beta"]), g["dx"], float(g["dt"]),
                       alpha_floodplains=g["alpha2"], components=(128, 128))
    assert kw.graph.components["tiers"] >= 2
    Q1, Q2 = g["Q0"].copy(), g["Q0_2"].copy()
    for s in range(g["q"].shape[0]):
        kw.kinematicWaveRouting(Q1, g["q"][s], "main_channel")
        kw.kinematicWaveRouting(Q2, 0.25 * g["q"][s], "floodplains")
        np.testing.assert_allclose(Q1, g["Q"][s], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(Q2, g["Q_2"][s], rtol=1e-9, atol=1e-12)
    assert kw.last_launches()["launches"] <= kw.graph.components["tiers"] + 1


@pytest.mark.parametrize("family,shape,comp", [("deep", (700, 500), (512, 512)), ("shallow", (600, 800), (2048, 4096)),
                                               ("deep", (3000, 64), (64, 64)), ("saddle", (300, 300), True)])
def test_components_equal_the_level_sweep(amd, solver, family, shape, comp):
    from lisflood_amd import synthetic as syn
    from lisflood_amd._lib import DeviceArray
    from lisflood_amd.kinematic_wave_parallel import kinematicWave
    H, W = shape
    mask = np.ones((H, W), bool)
    if family == "saddle":
        mask[:40, :30] = False
    codes = syn.make_ldd(family, H, W, 3, land_mask=mask)[mask].astype(np.float64)
    N = codes.size
    p = syn.router_params(N, seed=8)
    a = kinematicWave(codes, mask, p["alpha"], p["beta"], p["dx"], p["dt"])
    b = kinematicWave(codes, mask, p["alpha"], p["beta"], p["dx"], p["dt"], components=comp)
    Qa, Qb = p["Q0"].copy(), p["Q0"].copy()
    for s in range(3):                                      # pixel-order host call
        q = syn.lateral_inflow(N, s)
        a.kinematicWaveRouting(Qa, q)
        b.kinematicWaveRouting(Qb, q)
        assert np.array_equal(Qa, Qb), s
    assert np.isfinite(Qa).all() and Qa.max() > 0
    # engine-order resident call: each router in its own order
    perm_b = b.graph.layout()[0].astype(np.int64)
    dq = DeviceArray.from_host(np.ascontiguousarray(Qb[perm_b]))
    for s in range(3, 6):
        q = syn.lateral_inflow(N, s)
        a.kinematicWaveRouting(Qa, q)
        dl = DeviceArray.from_host(np.ascontiguousarray(q[perm_b]))
        b.route_ordered(dq, dl)
        dl.free()
    out = np.empty(N); out[perm_b] = dq.download()
    assert np.array_equal(out, Qa)
    # LDD reductions on the same layout
    w = np.random.default_rng(1).uniform(0, 10, N)
    assert np.array_equal(a.upstream_sum(w), b.upstream_sum(w))
    assert np.array_equal(a.accuflux(w), b.accuflux(w))
    st = b.graph.components
    assert b.last_launches()["launches"] <= st["tiers"] + 1 < a.graph.num_levels + 2 or family == "shallow"
    dq.free(); a.close(); b.close()


def test_components_scalar_dx_and_substep_module(amd):
    """scalar dx, and a whole routing sub-step (pixel order) whose two router calls run on the component layout"""
    from lisflood_amd import routing as RT
    g = golden("substep_split")
    v_a, v_b = RT.var_from_fixture(g), RT.var_from_fixture(g)
    for v, comp in ((v_a, None), (v_b, (200, 200))):
        m = RT.routing(v, split_routing=True)
        m.attach_router(g["codes"], g["mask"], components=comp)
        for s in range(3):
            v.ToChanM3RunoffDt = g["ToChanM3RunoffDt"][s]
            m.dynamic(s)
    for k in ("ChanQKin", "Chan2QKin", "ChanM3Kin", "ChanQ", "sumDisDay", "CrossSection2Area"):
        assert np.array_equal(getattr(v_a, k), getattr(v_b, k)), k


@pytest.mark.parametrize("split", [True, False])
@pytest.mark.parametrize("family,shape,comp", [("deep", (500, 900), (24, 128)), ("shallow", (700, 700), True),
                                               ("river", (400, 500), (16, 64))])
def test_components_fused_substeps_equal_the_level_wavefront(amd, family, shape, comp, split):
    """A model step of NoRoutSteps sub-steps inside every bin of the component layout (k_comp_fused; the roots of a tier
    hand their router outputs of every sub-step to the next tier) against the level wavefront: every state and output
    vector bit for bit.  20 % non-channel pixels (isolated; inert ones are skipped by both)."""
    from lisflood_amd import synthetic as syn
    from lisflood_amd import ldd as L
    from lisflood_amd.kinematic_wave_parallel import kinematicWave
    from lisflood_amd.routing import _OUT, _STATE
    from bench_support import RoutingStepDevice
    H, W = shape
    N = H * W
    mask = np.ones((H, W), bool)
    codes = syn.make_ldd(family, H, W, 5).reshape(-1).astype(np.float64)
    rng = np.random.default_rng(23)
    is_chan = rng.random(N) < 0.8
    kin, _ = L.lddmask(codes, mask, is_chan)
    ldd_kin = np.zeros(N); ldd_kin[is_chan] = kin
    p = syn.router_params(N, seed=12)
    beta, dt, nsteps = p["beta"], 3600.0, 9
    alpha, length = p["alpha"], p["dx"]
    alpha2 = alpha * rng.uniform(1.2, 2.0, N)
    q0 = np.where(is_chan, p["Q0"], 0.0)
    qlimit = 2.0 * q0 * rng.uniform(0.3, 1.2, N)
    vals = dict(ChanLength=length, InvChanLength=1 / length, ChannelAlpha=alpha, InvChannelAlpha=1 / alpha,
                ChannelAlpha2=alpha2, InvChannelAlpha2=1 / alpha2, QLimit=qlimit, M3Limit=alpha * length * qlimit ** beta,
                Chan2M3Start=alpha2 * length * qlimit ** beta, Chan2QStart=qlimit * 0.1, PixelArea=np.full(N, 2.5e7),
                IsChannelKinematic=is_chan, SideflowChanM3=syn.lateral_inflow(N, 0) * length * dt)
    vals["Chan2M3Kin"] = vals["Chan2M3Start"].copy()
    vals["ChanM3Kin"] = alpha * length * q0 ** beta
    vals["ChanQKin"] = q0.copy()
    vals["Chan2QKin"] = (vals["Chan2M3Kin"] / length / alpha2) ** (1 / beta)
    kwa = kinematicWave(ldd_kin, mask, alpha, beta, length, dt, alpha_floodplains=alpha2)
    kwb = kinematicWave(ldd_kin, mask, alpha, beta, length, dt, alpha_floodplains=alpha2, components=comp)
    a = RoutingStepDevice(kwa, vals, split, beta, 1 / dt, dt * nsteps)
    b = RoutingStepDevice(kwb, vals, split, beta, 1 / dt, dt * nsteps)
    for rep in range(2):                                   # two model steps: the second starts from the first's state
        a.run_fused(nsteps); b.run_fused(nsteps)
        names = _STATE + _OUT if split else ["ChanQKin", "ChanM3Kin", "ChanQ", "sumDisDay"] + _OUT
        for k in names:
            assert np.array_equal(a.download(k), b.download(k), equal_nan=True), (family, rep, k)
    q = b.download("ChanQ")
    assert np.isfinite(q).all() and q[is_chan].max() > 0
    assert kwb.last_launches()["launches"] <= kwb.graph.components["tiers"] + 1
    if family != "shallow":
        assert kwb.graph.components["tiers"] >= 2 and kwb.graph.components["trunk_cells"] > 1000   # root slabs in use
    a.free(); b.free(); kwa.close(); kwb.close()
