"""Row-block partitioned kinematic-wave routing (one process per GPU, RCCL halo exchange over xGMI).

The reference has no distributed code; this module is the host side of csrc/lf_dist.hip.  A transport is
any object with `exchange_int32(top_send, bottom_send, n_top_recv, n_bottom_recv) -> (top_recv, bottom_recv)`
and `allreduce_max(int) -> int`: `SocketTransport` (plain TCP, no PyTorch -- what bench.py uses; a torch.distributed
transport with the same two methods lives in tests/dist_worker_gloo.py); `settle_phases_local` / `loopback_route` connect
several blocks living in one process (tests, single-GPU loopback).
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import DeviceArray, check, f64, lib, ptr, u8


def row_blocks(H, nranks):
    """Contiguous row ranges [(r0, r1)] of an H-row raster, as even as possible."""
    base, extra = divmod(H, nranks)
    out, r = [], 0
    for k in range(nranks):
        n = base + (1 if k < extra else 0)
        out.append((r, r + n))
        r += n
    return out


class DistGraph:
    """One rank's block.  ldd_local [Hl, W] uint8; top / bottom: the halo row (W uint8) or None."""

    def __init__(self, ldd_local, mask_local=None, ldd_top=None, mask_top=None, ldd_bottom=None, mask_bottom=None):
        ldd_local = np.ascontiguousarray(ldd_local, dtype=np.uint8)
        Hl, W = ldd_local.shape
        self.shape = (Hl, W)
        conv = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.uint8)
        keep = [ldd_local, None if mask_local is None else u8(mask_local), conv(ldd_top),
                None if mask_top is None else u8(mask_top), conv(ldd_bottom),
                None if mask_bottom is None else u8(mask_bottom)]
        self._h = C.c_void_p()
        check(lib().lf_dist_graph_create(ptr(keep[0]), ptr(keep[1]), C.c_int(Hl), C.c_int(W), ptr(keep[2]),
                                         ptr(keep[3]), ptr(keep[4]), ptr(keep[5]), C.byref(self._h)))
        L = lib()
        L.lf_dist_graph_num_pixels.restype = C.c_int64
        L.lf_dist_graph_state_size.restype = C.c_int64
        L.lf_dist_graph_num_launch_units.restype = C.c_int64
        L.lf_dist_graph_num_noncontiguous.restype = C.c_int64
        L.lf_dist_graph_round_recv_slot.restype = C.c_int64
        self.num_pixels = int(L.lf_dist_graph_num_pixels(self._h))
        c = (C.c_int64 * 4)()
        check(L.lf_dist_graph_counts(self._h, c))
        self.n_export = (int(c[0]), int(c[1]))
        self.n_ghost = (int(c[2]), int(c[3]))
        self.finalized = False

    # --- phase fixpoint ---------------------------------------------------------------------------
    def export_phases(self):
        t = np.zeros(self.n_export[0], np.int32)
        b = np.zeros(self.n_export[1], np.int32)
        check(lib().lf_dist_graph_get_export_phases(self._h, ptr(t), ptr(b)))
        return t, b

    def set_ghost_phases(self, top, bottom):
        top = np.ascontiguousarray(top, dtype=np.int32)
        bottom = np.ascontiguousarray(bottom, dtype=np.int32)
        assert top.size == self.n_ghost[0] and bottom.size == self.n_ghost[1]
        ch = C.c_int(0)
        check(lib().lf_dist_graph_set_ghost_phases(self._h, ptr(top), ptr(bottom), C.byref(ch)))
        return bool(ch.value)

    def local_num_phases(self):
        return int(lib().lf_dist_graph_local_num_phases(self._h))

    def finalize(self, nphases):
        check(lib().lf_dist_graph_finalize(self._h, C.c_int(nphases)))
        self.finalized = True
        self.num_phases = nphases
        self.state_size = int(lib().lf_dist_graph_state_size(self._h))
        self.num_launch_units = int(lib().lf_dist_graph_num_launch_units(self._h))
        self.num_noncontiguous = int(lib().lf_dist_graph_num_noncontiguous(self._h))

    # --- plan getters -----------------------------------------------------------------------------
    def layout(self):
        perm = np.empty(self.num_pixels, np.int32)
        ph = np.empty(self.num_pixels, np.int32)
        check(lib().lf_dist_graph_get_layout(self._h, ptr(perm), ptr(ph)))
        return perm, ph

    def csr(self):
        ne = C.c_int64(0)
        check(lib().lf_dist_graph_get_csr(self._h, None, None, C.byref(ne)))
        ups_ptr = np.empty(self.num_pixels + 1, np.int32)
        ups_idx = np.empty(ne.value, np.int32)
        check(lib().lf_dist_graph_get_csr(self._h, ptr(ups_ptr), ptr(ups_idx), C.byref(ne)))
        return ups_ptr, ups_idx

    def phase_range(self, phase):
        o = (C.c_int64 * 2)()
        check(lib().lf_dist_graph_phase_range(self._h, C.c_int(phase), o))
        return int(o[0]), int(o[1])

    def part_range(self, phase, part):
        """positions [begin, end) of a phase's boundary-critical cells (part 0: its exports and what drains into them
        inside the phase) or of the rest (part 1); the two parts are independent of each other"""
        o = (C.c_int64 * 2)()
        check(lib().lf_dist_graph_part_range(self._h, C.c_int(phase), C.c_int(part), o))
        return int(o[0]), int(o[1])

    def round_counts(self, rnd):
        o = (C.c_int64 * 4)()
        check(lib().lf_dist_graph_round_counts(self._h, C.c_int(rnd), o))
        return dict(send=(int(o[0]), int(o[1])), recv=(int(o[2]), int(o[3])))

    def round_send_positions(self, rnd, side):
        n = self.round_counts(rnd)["send"][side]
        pos = np.empty(n, np.int32)
        check(lib().lf_dist_graph_round_send_positions(self._h, C.c_int(rnd), C.c_int(side), ptr(pos)))
        return pos

    def round_recv_slot(self, rnd, side):
        return int(lib().lf_dist_graph_round_recv_slot(self._h, C.c_int(rnd), C.c_int(side)))

    def slab_layout(self):
        """slab slots of the fused sub-step path: dict(slots, export=(top, bottom), ghost=(top, bottom), xphase)"""
        o = (C.c_int64 * 6)()
        check(lib().lf_dist_graph_slab_layout(self._h, o))
        return dict(slots=int(o[0]), export=(int(o[1]), int(o[2])), ghost=(int(o[3]), int(o[4])), xphase=int(o[5]))

    def route_plan(self):
        """the block plan of single router calls (lf_dist_graph_get_route_plan): dict(stage_block, level, row, off, cone,
        level_start) or None when no block holds more than one launch unit"""
        sizes = (C.c_int64 * 5)()
        check(lib().lf_dist_graph_get_route_plan(self._h, sizes, None, None, None, None, None, None))
        ls = np.empty(int(sizes[4]), np.int64)
        if sizes[0] == 0:
            check(lib().lf_dist_graph_get_route_plan(self._h, sizes, None, None, None, None, None, ptr(ls)))
            return None
        sb, lv, row = (np.empty(int(sizes[i]), np.int32) for i in range(3))
        off = np.empty(int(sizes[1]) - 1, np.int32)
        cone = np.empty(int(sizes[3]), np.int32)
        check(lib().lf_dist_graph_get_route_plan(self._h, sizes, ptr(sb), ptr(lv), ptr(row), ptr(off), ptr(cone), ptr(ls)))
        return dict(stage_block=sb, level=lv, row=row, off=off, cone=cone, level_start=ls)

    def fused_plan(self):
        """the block plan of the fused sub-step path, one plan per PHASE (lf_dist_graph_get_fused_plan), in the layout of
        route_plan() with `stage_block` = first block of every phase; None when no block holds more than one unit"""
        sizes = (C.c_int64 * 4)()
        check(lib().lf_dist_graph_get_fused_plan(self._h, sizes, None, None, None, None, None))
        if sizes[0] == 0:
            return None
        pb, lv, row = (np.empty(int(sizes[i]), np.int32) for i in range(3))
        off = np.empty(int(sizes[1]) - 1, np.int32)
        cone = np.empty(int(sizes[3]), np.int32)
        check(lib().lf_dist_graph_get_fused_plan(self._h, sizes, ptr(pb), ptr(lv), ptr(row), ptr(off), ptr(cone)))
        rp_sizes = (C.c_int64 * 5)()
        check(lib().lf_dist_graph_get_route_plan(self._h, rp_sizes, None, None, None, None, None, None))
        ls = np.empty(int(rp_sizes[4]), np.int64)
        check(lib().lf_dist_graph_get_route_plan(self._h, rp_sizes, None, None, None, None, None, ptr(ls)))
        return dict(stage_block=pb, level=lv, row=row, off=off, cone=cone, level_start=ls)

    def fused_tables(self):
        """(out_slot[N] by position, ups_idx_f[n_edges]): see lf_dist_graph_get_fused_tables"""
        out_slot = np.empty(self.num_pixels, np.int32)
        idx = np.empty(self.csr()[1].size, np.int32)
        check(lib().lf_dist_graph_get_fused_tables(self._h, ptr(out_slot), ptr(idx)))
        return out_slot, idx

    def close(self):
        if self._h:
            lib().lf_dist_graph_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def settle_phases(graph, transport):
    """Distributed fixpoint of the phase numbers, then finalize with the global phase count."""
    while True:
        t, b = graph.export_phases()
        gt, gb = transport.exchange_int32(t, b, graph.n_ghost[0], graph.n_ghost[1])
        changed = graph.set_ghost_phases(gt, gb)
        if not transport.allreduce_max(1 if changed else 0):
            break
    graph.finalize(transport.allreduce_max(graph.local_num_phases()))
    return graph


def settle_phases_local(graphs):
    """Same fixpoint as settle_phases for blocks held in one process (rank k is graphs[k])."""
    R = len(graphs)
    while True:
        exports = [g.export_phases() for g in graphs]
        changed = False
        for k, g in enumerate(graphs):
            gt = exports[k - 1][1] if k > 0 else np.zeros(0, np.int32)       # my top ghosts = upper rank's bottom exports
            gb = exports[k + 1][0] if k + 1 < R else np.zeros(0, np.int32)   # my bottom ghosts = lower rank's top exports
            changed |= g.set_ghost_phases(gt, gb)
        if not changed:
            break
    nph = max(g.local_num_phases() for g in graphs)
    for g in graphs:
        g.finalize(nph)
    return nph


# ---- wire format of SocketTransport: a fixed, typed encoding (no pickle -- nothing a peer sends is ever executed) ----------
#   N                      None
#   i <int64>              int            f <float64>   float          T / F   bool
#   b <u64 n> <n bytes>    bytes          s <u64 n> <utf-8>  str
#   a <u8 len><dtype str> <u8 ndim> <u64 shape...> <raw C-order bytes>    numpy array (numeric dtypes only)
#   l <u64 n> items / t <u64 n> items                                     list / tuple
_NUMERIC_KINDS = "biuf"


def _encode(obj, out):
    import struct
    if obj is None:
        out.append(b"N")
    elif isinstance(obj, (bool, np.bool_)):
        out.append(b"T" if obj else b"F")
    elif isinstance(obj, (int, np.integer)):
        out.append(b"i" + struct.pack("<q", int(obj)))
    elif isinstance(obj, (float, np.floating)):
        out.append(b"f" + struct.pack("<d", float(obj)))
    elif isinstance(obj, (bytes, bytearray)):
        out.append(b"b" + struct.pack("<Q", len(obj)) + bytes(obj))
    elif isinstance(obj, str):
        raw = obj.encode("utf-8")
        out.append(b"s" + struct.pack("<Q", len(raw)) + raw)
    elif isinstance(obj, np.ndarray):
        if obj.dtype.kind not in _NUMERIC_KINDS:
            raise TypeError("SocketTransport sends numeric arrays only, not dtype %r" % (obj.dtype,))
        a = np.ascontiguousarray(obj)
        dt = a.dtype.str.encode("ascii")
        out.append(b"a" + struct.pack("<B", len(dt)) + dt + struct.pack("<B", a.ndim) +
                   struct.pack("<%dQ" % a.ndim, *a.shape) + a.tobytes())
    elif isinstance(obj, (list, tuple)):
        out.append((b"l" if isinstance(obj, list) else b"t") + struct.pack("<Q", len(obj)))
        for x in obj:
            _encode(x, out)
    else:
        raise TypeError("SocketTransport cannot send %r" % (type(obj),))


def _decode(buf, pos=0, depth=0):
    """-> (object, next position); raises ValueError on anything malformed"""
    import struct
    if depth > 16:
        raise ValueError("message nested too deeply")
    tag = buf[pos:pos + 1]
    pos += 1
    if tag == b"N":
        return None, pos
    if tag in (b"T", b"F"):
        return tag == b"T", pos
    if tag == b"i":
        return struct.unpack_from("<q", buf, pos)[0], pos + 8
    if tag == b"f":
        return struct.unpack_from("<d", buf, pos)[0], pos + 8
    if tag in (b"b", b"s"):
        n, = struct.unpack_from("<Q", buf, pos)
        pos += 8
        if n > len(buf) - pos:
            raise ValueError("truncated message")
        raw = bytes(buf[pos:pos + n])
        return (raw if tag == b"b" else raw.decode("utf-8")), pos + n
    if tag == b"a":
        ln, = struct.unpack_from("<B", buf, pos)
        dt = np.dtype(bytes(buf[pos + 1:pos + 1 + ln]).decode("ascii"))
        pos += 1 + ln
        if dt.kind not in _NUMERIC_KINDS:
            raise ValueError("refusing array dtype %r" % (dt,))
        nd, = struct.unpack_from("<B", buf, pos)
        shape = struct.unpack_from("<%dQ" % nd, buf, pos + 1)
        pos += 1 + 8 * nd
        count = 1
        for d in shape:
            count *= d
        nbytes = count * dt.itemsize
        if nbytes > len(buf) - pos:
            raise ValueError("truncated message")
        a = np.frombuffer(buf, dtype=dt, count=count, offset=pos).reshape(shape).copy()
        return a, pos + nbytes
    if tag in (b"l", b"t"):
        n, = struct.unpack_from("<Q", buf, pos)
        pos += 8
        if n > len(buf):
            raise ValueError("truncated message")
        items = []
        for _ in range(n):
            x, pos = _decode(buf, pos, depth + 1)
            items.append(x)
        return (items if tag == b"l" else tuple(items)), pos
    raise ValueError("unknown type tag %r" % (tag,))


def _private_dir():
    """a directory only this user can write: the rendezvous file must not be replaceable by another user of the node"""
    import stat
    import tempfile
    d = os.path.join(tempfile.gettempdir(), "lisflood_amd_%d" % os.getuid())
    try:
        os.mkdir(d, 0o700)
    except FileExistsError:
        pass
    st = os.lstat(d)
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise PermissionError("%s is not a private directory of this user" % d)
    return d


class SocketTransport:
    """Rendezvous and set-up transport over plain TCP sockets -- no PyTorch (BASELINE.json's north_star): rank 0 listens,
    every other rank connects, collectives are a gather to rank 0 and a scatter back.  Only set-up traffic goes through
    it (the phase fixpoint's int32 vectors, the 128-byte RCCL id, barriers and the max-over-ranks clock of the bench);
    the data path is RCCL.

    Rendezvous on one node: rank 0 binds an ephemeral port on 127.0.0.1 and publishes `port token` (a fresh random
    token per run, and its own pid) in a 0600 file inside a 0700 per-user directory, named after MASTER_ADDR / MASTER_PORT,
    the world size (+ torchrun's TORCHELASTIC_RUN_ID and LF_JOB_ID when present) -- so it works under `python -m torch.distributed.run`, whose agent owns
    MASTER_PORT itself, as well as under any launcher that sets RANK / WORLD_SIZE.  A peer must present the token before
    anything else it sends is looked at; a connection that does not is dropped.  Messages are a fixed typed encoding of
    None / bool / int / float / bytes / str / numeric numpy arrays / lists / tuples (`_encode`): nothing received is
    ever unpickled or evaluated.  A stale file of a crashed run (its rank 0's pid is gone) is removed by rank 0 before it
    binds, a file whose rank 0 is alive is a concurrent job's and is refused; the other ranks re-read the file and
    reconnect until the timeout, so reading a dead port first is harmless.  The handshake carries the world size."""

    _MAGIC = b"LFAMD1"
    _MAX_MESSAGE = 1 << 31

    def __init__(self, rank, nranks, rendezvous_file, timeout=300.0):
        import hmac
        import secrets
        import socket
        import struct
        import time
        self.rank, self.nranks, self._struct = rank, nranks, struct
        self.peers = {}
        if nranks == 1:
            return
        deadline = time.time() + timeout
        if rank == 0:
            try:                                    # a file left behind by a crashed run (its rank 0 is gone) is removed;
                with open(rendezvous_file) as f:    # one whose rank 0 is alive belongs to a concurrent job with the same
                    owner = int(f.read().split()[2])  # MASTER_ADDR / MASTER_PORT / world size: refuse, do not clobber it
                alive = owner != os.getpid()
                if alive:
                    try:
                        os.kill(owner, 0)
                    except ProcessLookupError:
                        alive = False
                    except OSError:
                        pass
                if alive:
                    raise RuntimeError("rendezvous file %s belongs to a running job (pid %d): give this job its own "
                                       "MASTER_PORT or LF_JOB_ID" % (rendezvous_file, owner))
            except (OSError, ValueError, IndexError):
                pass
            try:
                os.unlink(rendezvous_file)
            except OSError:
                pass
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.bind(("127.0.0.1", 0))
            srv.listen(max(nranks, 8))
            token = secrets.token_hex(16)
            tmp = rendezvous_file + ".tmp%d" % os.getpid()
            fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
            with os.fdopen(fd, "w") as f:
                f.write("%d %s %d" % (srv.getsockname()[1], token, os.getpid()))
            os.replace(tmp, rendezvous_file)
            try:
                while len(self.peers) < nranks - 1:
                    srv.settimeout(max(deadline - time.time(), 0.01))
                    try:
                        c, _addr = srv.accept()
                    except socket.timeout:
                        raise TimeoutError("only %d of %d ranks connected" % (len(self.peers) + 1, nranks))
                    try:                               # handshake: magic, token, rank, world size -- 6 + 32 + 4 + 4 bytes,
                        c.settimeout(1.0)              # sent at once by a real peer (an idle connection must not stall
                        hello = self._exact(c, len(self._MAGIC) + 32 + 8)     # the others for long)
                        r, w = struct.unpack("<ii", hello[-8:])
                        ok = (hello[:len(self._MAGIC)] == self._MAGIC and
                              hmac.compare_digest(hello[len(self._MAGIC):-8], token.encode("ascii")) and
                              w == nranks and 0 < r < nranks and r not in self.peers)
                        if ok:
                            c.sendall(b"OK")
                    except (OSError, ConnectionError):  # (a peer that dies mid-handshake is dropped, the loop goes on)
                        ok = False
                    if not ok:
                        c.close()                      # not one of ours
                        continue
                    c.settimeout(timeout)
                    c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    self.peers[r] = c
            finally:
                srv.close()
                try:
                    os.unlink(rendezvous_file)
                except OSError:
                    pass
        else:
            c = None
            while c is None:
                if time.time() > deadline:
                    raise TimeoutError("no rendezvous with rank 0 through %s" % rendezvous_file)
                try:
                    with open(rendezvous_file) as f:
                        port_s, token = f.read().split()[:2]
                    s = socket.create_connection(("127.0.0.1", int(port_s)), timeout=5.0)
                    try:
                        s.settimeout(10.0)
                        s.sendall(self._MAGIC + token.encode("ascii") + struct.pack("<ii", rank, nranks))
                        if self._exact(s, 2) != b"OK":
                            raise ConnectionError("handshake refused")
                        c = s
                    except (OSError, ConnectionError):
                        s.close()
                        raise
                except (OSError, ValueError, ConnectionError):    # no file yet, a stale one, a dead port: look again
                    time.sleep(0.05)
            c.settimeout(timeout)
            c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            self.peers[0] = c

    @classmethod
    def from_env(cls, timeout=300.0):
        """RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT as torch.distributed.run (or any launcher) exports them"""
        rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
        tag = "%s_%s_%s_%s_w%d" % (os.environ.get("MASTER_ADDR", "127.0.0.1"), os.environ.get("MASTER_PORT", "29500"),
                                   os.environ.get("TORCHELASTIC_RUN_ID", "none"), os.environ.get("LF_JOB_ID", "none"), world)
        name = "rdv_" + "".join(ch if ch.isalnum() else "_" for ch in tag)
        return cls(rank, world, os.path.join(_private_dir(), name) if world > 1 else name, timeout)

    @staticmethod
    def _exact(sock, n):
        buf = bytearray()
        while len(buf) < n:
            chunk = sock.recv(min(n - len(buf), 1 << 20))
            if not chunk:
                raise ConnectionError("peer closed the connection")
            buf += chunk
        return bytes(buf)

    def _send(self, sock, obj):
        parts = []
        _encode(obj, parts)
        data = b"".join(parts)
        sock.sendall(self._struct.pack("<q", len(data)) + data)

    def _recv(self, sock):
        n, = self._struct.unpack("<q", self._exact(sock, 8))
        if n < 1 or n > self._MAX_MESSAGE:
            raise ValueError("bad message length %d" % n)
        buf = self._exact(sock, n)
        obj, end = _decode(buf)
        if end != n:
            raise ValueError("trailing bytes in message")
        return obj

    def allgather(self, obj):
        """-> [obj of rank 0, obj of rank 1, ...] on every rank"""
        if self.nranks == 1:
            return [obj]
        if self.rank == 0:
            out = [obj] + [None] * (self.nranks - 1)
            for r, c in self.peers.items():
                out[r] = self._recv(c)
            for c in self.peers.values():
                self._send(c, out)
            return out
        self._send(self.peers[0], obj)
        return self._recv(self.peers[0])

    def barrier(self):
        self.allgather(None)

    def broadcast(self, obj, src=0):
        return self.allgather(obj if self.rank == src else None)[src]

    def allreduce(self, value, op="max"):
        vals = self.allgather(value)
        f = {"max": np.maximum, "min": np.minimum, "sum": np.add}[op]
        out = np.asarray(vals[0])
        for v in vals[1:]:
            out = f(out, np.asarray(v))
        return out

    def allreduce_max(self, value):
        return int(self.allreduce(int(value), "max"))

    def exchange_int32(self, top_send, bottom_send, n_top_recv, n_bottom_recv):
        """set-up vectors with the vertical neighbours (rank -/+ 1): my top ghosts = the upper rank's bottom exports"""
        allv = self.allgather((np.ascontiguousarray(top_send, np.int32), np.ascontiguousarray(bottom_send, np.int32)))
        top = allv[self.rank - 1][1] if self.rank > 0 else np.zeros(0, np.int32)
        bot = allv[self.rank + 1][0] if self.rank + 1 < self.nranks else np.zeros(0, np.int32)
        assert top.size == n_top_recv and bot.size == n_bottom_recv
        return top, bot

    def close(self):
        for c in self.peers.values():
            try:
                c.close()
            except OSError:
                pass
        self.peers = {}


class Comm:
    """RCCL communicator; `unique_id()` on rank 0, broadcast the bytes, then Comm(id, nranks, rank, device)."""

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        check(lib().lf_comm_unique_id(buf))
        return buf.raw

    def __init__(self, uid, nranks, rank, device):
        self._h = C.c_void_p()
        check(lib().lf_comm_create(C.c_char_p(uid), C.c_int(nranks), C.c_int(rank), C.c_int(device), C.byref(self._h)))

    def close(self):
        """tears the communicator down; raises if RCCL reports an error of an earlier (asynchronous) operation"""
        if self._h:
            h, self._h = self._h, C.c_void_p()
            check(lib().lf_comm_close(h))


class DistRouter:
    """One rank's share of a kinematicWave over the partitioned raster (device side)."""

    def __init__(self, graph, alpha, beta, space_delta, time_delta, alpha_floodplains=None, device=0, comm=None,
                 rank_top=-1, rank_bottom=-1):
        assert graph.finalized
        self.graph, self.device, self.comm = graph, device, comm
        self.rank_top, self.rank_bottom = rank_top, rank_bottom
        N = graph.num_pixels
        self.num_pixels = N
        alpha = f64(np.broadcast_to(alpha, (N,)))
        if np.ndim(space_delta) == 0:
            dx, dxs = None, float(space_delta)
        else:
            dx, dxs = f64(space_delta), 0.0
        a2 = None if alpha_floodplains is None else f64(np.broadcast_to(alpha_floodplains, (N,)))
        self._h = C.c_void_p()
        check(lib().lf_dist_router_create(graph._h, ptr(alpha), C.c_double(beta), ptr(dx), C.c_double(dxs),
                                          C.c_double(time_delta), ptr(a2), C.c_int(device), C.byref(self._h)))
        lib().lf_dist_router_state_size.restype = C.c_int64
        lib().lf_dist_router_last_launches.restype = C.c_int64
        self.state_size = graph.state_size

    def new_state(self, pix_values=None):
        """Device state vector (local cells in engine order + ghost slots), optionally initialised from a
        local-pixel-order host vector."""
        st = DeviceArray(max(self.state_size, 1), np.float64, self.device).zero()
        if pix_values is not None:
            tmp = DeviceArray.from_host(f64(pix_values), self.device)
            check(lib().lf_dist_router_to_engine_order(self._h, tmp.ptr, st.ptr))
            _lib.synchronize(self.device)
            tmp.free()
        return st

    def download_pix(self, state):
        tmp = DeviceArray(max(self.num_pixels, 1), np.float64, self.device)
        check(lib().lf_dist_router_from_engine_order(self._h, state.ptr, tmp.ptr))
        out = tmp.download()[:self.num_pixels]
        tmp.free()
        return out

    def route(self, q_state, lat_state, section="main_channel"):
        sec = _lib.SECTION[section]
        ch = self.comm._h if self.comm is not None else None
        check(lib().lf_dist_router_route(self._h, ch, q_state.ptr, lat_state.ptr, C.c_int(sec), C.c_int(self.rank_top),
                                         C.c_int(self.rank_bottom)))

    def route_many(self, q_state, lat_states, section="main_channel"):
        """len(lat_states) calls in a row, pipelined across calls (lf_dist_router_route_many); result in q_state"""
        sec = _lib.SECTION[section]
        ch = self.comm._h if self.comm is not None else None
        n = len(lat_states)
        arr = (C.c_void_p * max(n, 1))(*[d.ptr.value for d in lat_states])
        check(lib().lf_dist_router_route_many(self._h, ch, q_state.ptr, arr, C.c_int(n), C.c_int(sec),
                                              C.c_int(self.rank_top), C.c_int(self.rank_bottom)))

    def compute_part_io(self, q_in, q_out, lat_state, phase, part, section="main_channel"):
        check(lib().lf_dist_router_compute_part_io(self._h, q_in.ptr, q_out.ptr, lat_state.ptr,
                                                   C.c_int(_lib.SECTION[section]), C.c_int(phase), C.c_int(part)))

    # pieces, for the in-process loopback
    def compute_phase(self, q_state, lat_state, phase, section="main_channel"):
        check(lib().lf_dist_router_compute_phase(self._h, q_state.ptr, lat_state.ptr, C.c_int(_lib.SECTION[section]),
                                                 C.c_int(phase)))

    def compute_part(self, q_state, lat_state, phase, part, section="main_channel"):
        check(lib().lf_dist_router_compute_part(self._h, q_state.ptr, lat_state.ptr, C.c_int(_lib.SECTION[section]),
                                                C.c_int(phase), C.c_int(part)))

    def pack(self, q_state, rnd):
        ptrs = (C.c_void_p * 2)()
        cnt = (C.c_int64 * 2)()
        check(lib().lf_dist_router_pack(self._h, q_state.ptr, C.c_int(rnd), ptrs, cnt))
        return [(ptrs[i], int(cnt[i])) for i in range(2)]

    def exchange(self, q_state, rnd):
        """halo round `rnd` of a router call alone (lf_dist_router_exchange: pack, one grouped RCCL Send/Recv per neighbour
        on the library stream) -- what route() issues after part 0 of phase `rnd`"""
        ch = self.comm._h if self.comm is not None else None
        check(lib().lf_dist_router_exchange(self._h, ch, q_state.ptr, C.c_int(rnd), C.c_int(self.rank_top),
                                            C.c_int(self.rank_bottom)))

    def recv_slots(self, rnd):
        slot = (C.c_int64 * 2)()
        cnt = (C.c_int64 * 2)()
        check(lib().lf_dist_router_recv_slots(self._h, C.c_int(rnd), slot, cnt))
        return [(int(slot[i]), int(cnt[i])) for i in range(2)]

    def last_launches(self):
        return int(lib().lf_dist_router_last_launches(self._h))

    def close(self):
        if getattr(self, "_h", None):
            lib().lf_dist_router_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DistRoutingStep:
    """One rank's vectors of routing.dynamic() (routing.py:435-706) on the partition, resident in the rank's engine
    order.  `substep()` = lf_dist_routing_substep (RCCL halo inside each router call); `stage(i)` runs one of the
    element-wise stages alone (the in-process loopback drives the router calls itself)."""

    def __init__(self, router, values, split, Beta, InvDtRouting, DtSec):
        """values: name -> host vector of the rank's cells in local pixel order (names of lf_substep_args)."""
        from .routing import _OUT, _STATE, _STATIC, _SubstepArgs
        self.router, self.split, self.device = router, bool(split), router.device
        N = self.N = router.num_pixels
        self.perm = router.graph.layout()[0].astype(np.int64)
        self.dev = {}
        zeros = np.zeros(N)
        for k in _STATIC + _STATE:
            x = values.get(k)
            if x is None:
                x = np.ones(N, bool) if k == "IsChannelKinematic" else zeros
            x = np.broadcast_to(x, (N,))[self.perm]
            if k in ("ChanQKin", "Chan2QKin"):          # router state vectors: ghost slots behind the N cells
                st = DeviceArray(max(router.state_size, 1), np.float64, self.device).zero()
                _upload_prefix(st, f64(x), self.device)
                self.dev[k] = st
            else:
                self.dev[k] = DeviceArray.from_host(_lib.u8(x) if k == "IsChannelKinematic" else f64(x), self.device)
        for k in _OUT + ["scratch0", "scratch1"]:
            self.dev[k] = DeviceArray(max(N, 1), np.float64, self.device).zero()
        side = np.broadcast_to(values.get("SideflowChanM3", zeros), (N,))[self.perm]
        self.dev["SideflowChanM3"] = DeviceArray.from_host(f64(side), self.device)
        a = self.args = _SubstepArgs()
        for k, d in self.dev.items():
            setattr(a, k, d.ptr.value)
        a.Beta, a.InvBeta, a.InvDtRouting, a.DtSec = float(Beta), 1.0 / float(Beta), float(InvDtRouting), float(DtSec)
        a.split, a.engine_order = (1 if self.split else 0), 1

    def substep(self):
        r = self.router
        ch = r.comm._h if r.comm is not None else None
        check(lib().lf_dist_routing_substep(r._h, ch, C.byref(self.args), C.c_int(r.rank_top), C.c_int(r.rank_bottom)))

    def stage(self, i):
        check(lib().lf_substep_stage(C.c_int(self.device), C.c_int(i), C.c_int64(self.N), C.byref(self.args)))

    # --- a whole model step: every sub-step of a phase as one wavefront, one halo exchange per phase -----------------
    def substeps_fused(self, nsteps):
        """nsteps x routing.dynamic() = lf_routing_substeps_fused on the whole raster (lf_dist_routing_substeps_fused)"""
        r = self.router
        ch = r.comm._h if r.comm is not None else None
        check(lib().lf_dist_routing_substeps_fused(r._h, ch, C.byref(self.args), C.c_int(nsteps), C.c_int64(0),
                                                   C.c_int(r.rank_top), C.c_int(r.rank_bottom)))

    def fused_prepare(self, nsteps):
        check(lib().lf_dist_fused_prepare(self.router._h, C.byref(self.args), C.c_int(nsteps)))

    def fused_phase(self, nsteps, phase):
        check(lib().lf_dist_fused_phase(self.router._h, C.byref(self.args), C.c_int(nsteps), C.c_int64(0), C.c_int(phase)))

    # --- several model steps per call: the sub-steps of all of them as one wavefront per phase, one halo block per phase ----
    def _model_step_args(self, sums, sideflows):
        from .routing import _SubstepArgs
        a = _SubstepArgs.from_buffer_copy(self.args)
        a.sumDisDay = sums.ptr.value
        stride = 0
        if sideflows is not None:
            a.SideflowChanM3 = sideflows.ptr.value
            stride = self.N
        return a, stride

    def model_steps_fused(self, nsteps, nmodel, sums, sideflows=None):
        """nmodel model steps of nsteps sub-steps in one call (lf_dist_routing_model_steps_fused).  sums: device array
        [nmodel, N] (zeroed by the caller) that receives every model step's discharge sum; sideflows: device array
        [nmodel, N] in the rank's engine order, or None for the resident vector in every model step."""
        r = self.router
        ch = r.comm._h if r.comm is not None else None
        a, stride = self._model_step_args(sums, sideflows)
        check(lib().lf_dist_routing_model_steps_fused(r._h, ch, C.byref(a), C.c_int(nsteps), C.c_int(nmodel), C.c_int64(stride),
                                                      C.c_int(r.rank_top), C.c_int(r.rank_bottom)))

    def fused_phase_model_steps(self, nsteps, nmodel, phase, sums, sideflows=None):
        a, stride = self._model_step_args(sums, sideflows)
        check(lib().lf_dist_fused_phase_model_steps(self.router._h, C.byref(a), C.c_int(nsteps), C.c_int(nmodel),
                                                    C.c_int64(stride), C.c_int(phase)))

    def fused_halo_block(self, rnd, side):
        """(send offset, send count, recv offset, recv count) in doubles inside the slab of a section"""
        o = (C.c_int64 * 4)()
        check(lib().lf_dist_fused_halo_block(self.router._h, C.c_int(rnd), C.c_int(side), o))
        return tuple(int(x) for x in o)

    def fused_slab(self, section):
        p = C.c_void_p()
        check(lib().lf_dist_fused_slab(self.router._h, C.c_int(section), C.byref(p)))
        return p.value or 0

    def download(self, name):
        out = np.empty(self.N)
        out[self.perm] = self.dev[name].download()[:self.N]
        return out

    def free(self):
        for d in self.dev.values():
            d.free()
        self.dev = {}


def _upload_prefix(dst, host, device):
    """host vector -> the first host.size entries of a larger device array"""
    if host.size:
        check(lib().lf_memcpy_h2d(C.c_int(device), dst.ptr, ptr(host), C.c_size_t(host.nbytes)))


def loopback_substep(steps):
    """One routing sub-step over blocks that all live on one GPU (see loopback_route)."""
    routers = [s.router for s in steps]
    for s in steps:
        s.stage(0)
    loopback_route(routers, [s.dev["ChanQKin"] for s in steps], [s.dev["scratch0"] for s in steps], "main_channel")
    for s in steps:
        s.stage(1)
    if steps[0].split:
        loopback_route(routers, [s.dev["Chan2QKin"] for s in steps], [s.dev["scratch1"] for s in steps], "floodplains")
        for s in steps:
            s.stage(2)


def loopback_substeps_fused(steps, nsteps, lanes=False):
    """A whole model step (nsteps sub-steps) over blocks that all live on one GPU: lf_dist_fused_phase per block and phase,
    the per-phase halo of the slabs as device-to-device copies -- the kernels and the plan of
    lf_dist_routing_substeps_fused, only the transport differs.  lanes: the blocks of a phase on separate streams (see
    loopback_model_steps_fused)."""
    dev = steps[0].device
    nph = steps[0].router.graph.num_phases
    L = lib()
    for s in steps:
        s.fused_prepare(nsteps)
    for j in range(nph):
        if lanes:
            check(L.lf_lane_fork(C.c_int(dev)))
        try:
            for k, s in enumerate(steps):
                if lanes:
                    check(L.lf_lane_select(C.c_int(dev), C.c_int(k + 1)))
                s.fused_phase(nsteps, j)
        finally:
            if lanes:
                check(L.lf_lane_select(C.c_int(dev), C.c_int(0)))
                check(L.lf_lane_join(C.c_int(dev)))
        if j + 1 < nph:
            _loopback_halo(steps, j)


def _loopback_halo(steps, j):
    """the slab blocks of halo round j between blocks that live on one GPU, as device-to-device copies"""
    R = len(steps)
    dev = steps[0].device
    for section in range(2 if steps[0].split else 1):
        blocks = [[s.fused_halo_block(j, side) for side in (0, 1)] for s in steps]
        slabs = [s.fused_slab(section) for s in steps]
        for k in range(R):
            for side, src_rank, src_side in ((0, k - 1, 1), (1, k + 1, 0)):
                _so, _sc, ro, rc = blocks[k][side]
                if rc == 0:
                    continue
                so, sc, _ro, _rc = blocks[src_rank][src_side]
                assert sc == rc, (k, j, side, sc, rc)
                check(lib().lf_memcpy_d2d(C.c_int(dev), C.c_void_p(slabs[k] + 8 * ro),
                                          C.c_void_p(slabs[src_rank] + 8 * so), C.c_size_t(8 * rc)))


def loopback_model_steps_fused(steps, nsteps, nmodel, sums, sideflows=None, lanes=True):
    """`nmodel` model steps of nsteps sub-steps over blocks that all live on one GPU, all of them in ONE pass over the
    phases (lf_dist_fused_phase_model_steps per block and phase; the halo of a phase carries the slabs of every model
    step).  sums[k] / sideflows[k]: block k's [nmodel, N_k] device arrays (sideflows None: the resident vector).
    lanes: the blocks of a phase on separate streams (lf_lane_*): inside a phase they are independent -- each has its own GPU
    on real hardware --, so their launch chains overlap instead of queueing behind one another."""
    nph = steps[0].router.graph.num_phases
    dev = steps[0].device
    L = lib()
    for s in steps:
        s.fused_prepare(nsteps * nmodel)
    for j in range(nph):
        if lanes:
            check(L.lf_lane_fork(C.c_int(dev)))
        try:
            for k, s in enumerate(steps):
                if lanes:
                    check(L.lf_lane_select(C.c_int(dev), C.c_int(k + 1)))
                s.fused_phase_model_steps(nsteps, nmodel, j, sums[k], None if sideflows is None else sideflows[k])
        finally:
            if lanes:
                check(L.lf_lane_select(C.c_int(dev), C.c_int(0)))
                check(L.lf_lane_join(C.c_int(dev)))
        if j + 1 < nph:
            _loopback_halo(steps, j)


def loopback_route_many(routers, q_states, lat_lists, late_halo=False, section="main_channel"):
    """lf_dist_router_route_many's schedule over blocks that live on one GPU: the same kernels in the same order on the
    two alternating state vectors, every halo round as a device copy -- performed right where the communication stream
    could first run it (late_halo=False) or right where the compute stream waits for it (late_halo=True): the two ends of
    the window in which the real exchange may land.  lat_lists[k] = the lateral inflow vectors of block k, one per call.
    The result ends in q_states (as the C function leaves it)."""
    R = len(routers)
    P = routers[0].graph.num_phases
    K = len(lat_lists[0])
    dev = routers[0].device
    if P < 2 or K < 2:
        for c in range(K):
            loopback_route(routers, q_states, [l[c] for l in lat_lists], section)
        return
    second = [DeviceArray(max(r.state_size, 1), np.float64, dev).zero() for r in routers]
    B = [q_states, second]
    pending = {}

    def part(c, j, pt):
        for k in range(R):
            routers[k].compute_part_io(B[c & 1][k], B[(c + 1) & 1][k], lat_lists[k][c], j, pt, section)

    def do_round(c, j):
        out = B[(c + 1) & 1]
        sends = [routers[k].pack(out[k], j) for k in range(R)]
        for k in range(R):
            slots = routers[k].recv_slots(j)
            for side, src_rank, src_side in ((0, k - 1, 1), (1, k + 1, 0)):
                slot, n = slots[side]
                if n == 0:
                    continue
                sp, sn = sends[src_rank][src_side]
                assert sn == n
                check(lib().lf_memcpy_d2d(C.c_int(dev), C.c_void_p(out[k].ptr.value + 8 * slot), C.c_void_p(sp),
                                          C.c_size_t(8 * n)))

    def issue_round(c, j):
        if j + 1 >= P:
            return
        if late_halo:
            pending[(c, j)] = True
        else:
            do_round(c, j)

    def wait_round(c, j):
        if pending.pop((c, j), False):
            do_round(c, j)

    part(0, 0, 0); issue_round(0, 0); part(0, 0, 1)
    for c in range(K):
        nxt = c + 1 < K
        if nxt:
            part(c + 1, 0, 0); issue_round(c + 1, 0)
        for j in range(1, P):
            wait_round(c, j - 1)
            part(c, j, 0); issue_round(c, j)
            if j == 1 and nxt:
                part(c + 1, 0, 1)
            part(c, j, 1)
    if K & 1:
        for k in range(R):
            n = routers[k].num_pixels
            if n:
                check(lib().lf_memcpy_d2d(C.c_int(dev), q_states[k].ptr, second[k].ptr, C.c_size_t(8 * n)))
    _lib.synchronize(dev)
    for d in second:
        d.free()


def loopback_route(routers, q_states, lat_states, section="main_channel", overlap_order=False):
    """One call over blocks that all live on ONE GPU in ONE process: the halo exchange is a device-to-device
    copy instead of RCCL Send/Recv.  Exercises exactly the kernels and the plan of the multi-GPU path.
    overlap_order: the order lf_dist_router_route's two streams allow -- a phase's boundary-critical part, the round's
    packs, then the bulk part, then the copies into the ghost slots."""
    R = len(routers)
    nph = routers[0].graph.num_phases
    dev = routers[0].device
    for j in range(nph):
        sends = None
        if overlap_order and j + 1 < nph:
            for k in range(R):
                routers[k].compute_part(q_states[k], lat_states[k], j, 0, section)
            sends = [routers[k].pack(q_states[k], j) for k in range(R)]
            for k in range(R):
                routers[k].compute_part(q_states[k], lat_states[k], j, 1, section)
        else:
            for k in range(R):
                routers[k].compute_phase(q_states[k], lat_states[k], j, section)
        if j + 1 < nph:
            if sends is None:
                sends = [routers[k].pack(q_states[k], j) for k in range(R)]
            for k in range(R):
                slots = routers[k].recv_slots(j)
                # side 0: from the rank above (its bottom send buffer); side 1: from the rank below (its top buffer)
                for side, src_rank, src_side in ((0, k - 1, 1), (1, k + 1, 0)):
                    slot, n = slots[side]
                    if n == 0:
                        continue
                    sp, sn = sends[src_rank][src_side]
                    assert sn == n, (k, j, side, sn, n)
                    dst = C.c_void_p(q_states[k].ptr.value + 8 * slot)
                    check(lib().lf_memcpy_d2d(C.c_int(dev), dst, C.c_void_p(sp), C.c_size_t(8 * n)))
