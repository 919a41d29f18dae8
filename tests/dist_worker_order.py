"""Worker of tests/test_dist_multirank_gpu.py::test_a_misordered_rank_is_caught: every rank issues the halo rounds of one
router call through the product's lf_dist_router_exchange, in round order -- or, on the rank LF_TEST_REVERSED_RANK names,
in reverse.  The stand-in for RCCL records what was issued (FAKE_RCCL_LOG_DIR); the parent compares the ranks' logs."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lisflood-code_amd"))

from lisflood_amd import _lib, dist as D, synthetic as syn  # noqa: E402


def main():
    T = D.SocketTransport.from_env(timeout=120.0)
    rank, world = T.rank, T.nranks
    H, W = 300, 240
    codes = syn.make_ldd(os.environ.get("LF_TEST_FAMILY", "saddle"), H, W, 6)
    N = H * W
    p = syn.router_params(N, seed=9)
    r0, r1 = D.row_blocks(H, world)[rank]
    g = D.DistGraph(codes[r0:r1], None, codes[r0 - 1] if r0 > 0 else None, None, codes[r1] if r1 < H else None, None)
    D.settle_phases(g, T)
    comm = D.Comm(T.broadcast(D.Comm.unique_id() if rank == 0 else None), world, rank, 0)
    sel = slice(r0 * W, r1 * W)
    router = D.DistRouter(g, p["alpha"][sel], p["beta"], p["dx"][sel], p["dt"], device=0, comm=comm,
                          rank_top=rank - 1 if rank > 0 else -1, rank_bottom=rank + 1 if rank + 1 < world else -1)
    Q = router.new_state(p["Q0"][sel])
    rounds = list(range(g.num_phases - 1))
    assert len(rounds) >= 2, "the test needs a partition with several halo rounds"
    # the premise of the test: the rounds' message sizes are not the same read backwards (RCCL -- and the stand-in --
    # match messages of a pair by order and check only their size; a swap of equal-sized rounds moves data into the wrong
    # ghost slots without either noticing, which is why the product's order must be right by construction)
    sizes = [tuple(n for _, n in router.pack(Q, j)) for j in rounds]
    _lib.synchronize(0)
    asym = T.allgather(sizes != sizes[::-1])
    assert any(asym), "halo rounds of equal sizes on every rank: pick another raster"
    if os.environ.get("LF_TEST_REVERSED_RANK", "") == str(rank):
        rounds.reverse()
    ok = True
    try:
        for j in rounds:
            router.exchange(Q, j)
        _lib.synchronize(0)
    except Exception as e:          # the stand-in notices a count mismatch or times out
        print("exchange failed: %r" % (e,), file=sys.stderr, flush=True)
        ok = False
    try:
        comm.close()                # (writes the issue log)
    except Exception as e:
        print("communicator: %r" % (e,), file=sys.stderr, flush=True)
        ok = False
    if ok:
        print("ORDER_WORKER_OK rounds=%d" % len(rounds), flush=True)
    sys.exit(0 if ok else 3)


if __name__ == "__main__":
    main()
