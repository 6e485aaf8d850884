"""The shape of the C-ABI's collective (include/rrhip.h rr_comm_* /
rr_allgather_metric, csrc/comm.hip) for world sizes 2..8, checked WITHOUT a
GPU or RCCL: the library's RCCL entry points are replaced by recording
stand-ins through its test hook (rrdbg_comm_inject), and what must hold is
which collective is issued with which buffers, counts and roots -- equal
blocks one ncclAllGather, ragged blocks one group of per-root broadcasts at
rrmpg_amd.sharding.shard_bounds' offsets (the reference's loop over
parameter sets, rrmpg/tools/monte_carlo.py:61-71, cut into contiguous
blocks)."""

import ctypes

import pytest

from rrmpg_amd import _lib
from rrmpg_amd.sharding import shard_bounds


class _Uid(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_char * 128)]


_vp, _int, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
GET_ID = ctypes.CFUNCTYPE(_int, ctypes.POINTER(_Uid))
INIT = ctypes.CFUNCTYPE(_int, ctypes.POINTER(_vp), _int, _Uid, _int)
DESTROY = ctypes.CFUNCTYPE(_int, _vp)
BCAST = ctypes.CFUNCTYPE(_int, _vp, _vp, _sz, _int, _int, _vp, _vp)
GATHER = ctypes.CFUNCTYPE(_int, _vp, _vp, _sz, _int, _vp, _vp)
GROUP = ctypes.CFUNCTYPE(_int)
INIT_ALL = ctypes.CFUNCTYPE(_int, ctypes.POINTER(_vp), _int,
                            ctypes.POINTER(_int))


class FakeRccl:
    """Records every call; communicator handles are 0x1000 + rank."""

    def __init__(self):
        self.calls = []
        self.fail_gather = 0

        def get_id(p):
            p.contents.internal = b"rr-test-id"
            return 0

        def init(out, nranks, uid, rank):
            self.calls.append(("init", nranks, uid.internal, rank))
            out[0] = 0x1000 + rank
            return 0

        def destroy(comm):
            self.calls.append(("destroy", comm))
            return 0

        def bcast(send, recv, count, dtype, root, comm, stream):
            self.calls.append(("bcast", send, recv, count, dtype, root, comm,
                               stream))
            return 0

        def gather(send, recv, count, dtype, comm, stream):
            self.calls.append(("gather", send, recv, count, dtype, comm,
                               stream))
            return self.fail_gather

        def group(name):
            def fn():
                self.calls.append((name,))
                return 0
            return fn

        def init_all(out, ndev, devs):
            self.calls.append(("init_all", ndev,
                               [devs[j] for j in range(ndev)] if devs
                               else None))
            for j in range(ndev):
                out[j] = 0x2000 + j
            return 0

        self.keep = [GET_ID(get_id), INIT(init), DESTROY(destroy),
                     BCAST(bcast), GATHER(gather), GROUP(group("start")),
                     GROUP(group("end")), INIT_ALL(init_all)]
        self.table = (_vp * 8)(*[ctypes.cast(f, _vp) for f in self.keep])


@pytest.fixture()
def fake():
    lib = _lib.load()
    lib.rrdbg_comm_inject.restype = _int
    lib.rrdbg_comm_inject.argtypes = [ctypes.POINTER(_vp)]
    f = FakeRccl()
    assert lib.rrdbg_comm_inject(f.table) == 0
    yield lib, f
    assert lib.rrdbg_comm_inject(None) == 0


ALL, LOCAL, STREAM = 0x7000_0000, 0x5000_0000, 0x77


def _comm(lib, f, world, rank):
    ident = (ctypes.c_char * 128)()
    assert lib.rr_comm_unique_id(ident) == 0
    assert bytes(ident).startswith(b"rr-test-id")
    comm = _vp()
    assert lib.rr_comm_init(ctypes.byref(comm), world, rank, ident) == 0
    assert f.calls[-1][:2] == ("init", world) and f.calls[-1][3] == rank
    assert f.calls[-1][2] == b"rr-test-id"
    f.calls.clear()
    return comm


@pytest.mark.parametrize("world", [2, 3, 4, 5, 6, 7, 8])
def test_equal_blocks_are_one_allgather(fake, world):
    lib, f = fake
    n = 125_000 * world
    for rank in range(world):
        comm = _comm(lib, f, world, rank)
        a, b = shard_bounds(n, world, rank)
        # out of place, then in place (local = this rank's block of `all`)
        for local in (LOCAL, ALL + 8 * a):
            assert lib.rr_allgather_metric(comm, local, b - a, ALL, n,
                                           STREAM) == 0
            assert f.calls == [("gather", local, ALL, b - a, 8,
                                0x1000 + rank, STREAM)]
            f.calls.clear()
        assert lib.rr_comm_destroy(comm) == 0
        assert f.calls == [("destroy", 0x1000 + rank)]
        f.calls.clear()


@pytest.mark.parametrize("world", [2, 3, 4, 5, 6, 7, 8])
def test_ragged_blocks_are_one_group_of_per_root_broadcasts(fake, world):
    lib, f = fake
    n = 1_000_003 if 1_000_003 % world else 1_000_001
    assert n % world
    blocks = [shard_bounds(n, world, r) for r in range(world)]
    assert blocks[0][0] == 0 and blocks[-1][1] == n
    for rank in range(world):
        comm = _comm(lib, f, world, rank)
        a, b = blocks[rank]
        assert lib.rr_allgather_metric(comm, LOCAL, b - a, ALL, n,
                                       STREAM) == 0
        assert f.calls[0] == ("start",) and f.calls[-1] == ("end",)
        body = f.calls[1:-1]
        assert len(body) == world
        for root, call in enumerate(body):
            ra, rb = blocks[root]
            send = LOCAL if root == rank else ALL + 8 * ra
            assert call == ("bcast", send, ALL + 8 * ra, rb - ra, 8, root,
                            0x1000 + rank, STREAM), (rank, root)
        f.calls.clear()
        # a block of the wrong length never reaches the collective
        assert lib.rr_allgather_metric(comm, LOCAL, b - a + 1, ALL, n,
                                       STREAM) == -2
        assert b"holds" in lib.rr_last_error() and f.calls == []
        assert lib.rr_comm_destroy(comm) == 0
        f.calls.clear()


def test_fewer_sets_than_ranks_and_errors(fake):
    lib, f = fake
    # three scores over eight ranks: ranks 0..2 hold one, the others none --
    # only the non-empty blocks are broadcast, every rank takes part in each
    for rank in (0, 2, 5):
        comm = _comm(lib, f, 8, rank)
        a, b = shard_bounds(3, 8, rank)
        assert lib.rr_allgather_metric(comm, LOCAL if b > a else None, b - a,
                                       ALL, 3, STREAM) == 0
        roots = [c[5] for c in f.calls if c[0] == "bcast"]
        assert roots == [0, 1, 2]
        assert all(c[3] == 1 for c in f.calls if c[0] == "bcast")
        f.calls.clear()
        lib.rr_comm_destroy(comm)
    # zero scores over equal blocks: nothing to exchange
    comm = _comm(lib, f, 4, 1)
    assert lib.rr_allgather_metric(comm, None, 0, ALL, 0, STREAM) == 0
    assert f.calls == []
    # RCCL's error code comes back as RR_E_HIP with its text
    f.fail_gather = 5
    assert lib.rr_allgather_metric(comm, LOCAL, 10, ALL, 40, STREAM) == -3
    assert b"ncclAllGather" in lib.rr_last_error()
    f.fail_gather = 0
    # bad arguments
    assert lib.rr_allgather_metric(None, LOCAL, 10, ALL, 40, STREAM) == -1
    assert lib.rr_allgather_metric(comm, LOCAL, 10, None, 40, STREAM) == -1
    bad = _vp()
    ident = (ctypes.c_char * 128)()
    assert lib.rr_comm_init(ctypes.byref(bad), 4, 4, ident) == -2
    assert lib.rr_comm_init(ctypes.byref(bad), 0, 0, ident) == -2
    lib.rr_comm_destroy(comm)


@pytest.mark.parametrize("world", [1, 2, 4, 8])
@pytest.mark.parametrize("ragged", [False, True])
def test_in_process_clique_one_thread_for_all_gpus(fake, world, ragged):
    """One process driving `world` GPUs (SURVEY.md 8e's sketch, what
    monte_carlo(sampler='device', gpus=G, exchange='rccl') does):
    rr_comm_init_all hands out the rank-j communicator of device devices[j],
    and the one thread issues the G collectives inside ONE group -- an
    all-gather per rank for equal blocks, a nested group of per-root
    broadcasts per rank for ragged ones, each with its own buffers."""
    lib, f = fake
    n = 1_000_000 + (3 if ragged and world > 1 else 0)
    if ragged and world == 1:
        pytest.skip("one rank has no ragged blocks")
    devs = (_int * world)(*[(3 + j) % 8 for j in range(world)])
    comms = (_vp * world)()
    assert lib.rr_comm_init_all(comms, world, devs) == 0, lib.rr_last_error()
    assert f.calls == [("init_all", world, [(3 + j) % 8 for j in range(world)])]
    assert [comms[j] for j in range(world)] != [None] * world
    f.calls.clear()
    assert lib.rr_comm_group_start() == 0
    for j in range(world):
        a, b = shard_bounds(n, world, j)
        assert lib.rr_allgather_metric(comms[j], LOCAL + 0x100000 * j, b - a,
                                       ALL + 0x1000000 * j, n,
                                       STREAM + j) == 0, lib.rr_last_error()
    assert lib.rr_comm_group_end() == 0
    assert f.calls[0] == ("start",) and f.calls[-1] == ("end",)
    body = f.calls[1:-1]
    if n % world == 0:
        assert body == [("gather", LOCAL + 0x100000 * j,
                         ALL + 0x1000000 * j, n // world, 8, 0x2000 + j,
                         STREAM + j) for j in range(world)]
    else:
        per_rank = world + 2              # nested start, W broadcasts, end
        assert len(body) == world * per_rank
        for j in range(world):
            mine = body[j * per_rank:(j + 1) * per_rank]
            assert mine[0] == ("start",) and mine[-1] == ("end",)
            for root, call in enumerate(mine[1:-1]):
                ra, rb = shard_bounds(n, world, root)
                send = (LOCAL + 0x100000 * j if root == j
                        else ALL + 0x1000000 * j + 8 * ra)
                assert call == ("bcast", send, ALL + 0x1000000 * j + 8 * ra,
                                rb - ra, 8, root, 0x2000 + j, STREAM + j)
    f.calls.clear()
    for j in range(world):
        assert lib.rr_comm_destroy(comms[j]) == 0
    assert [c[1] for c in f.calls] == [0x2000 + j for j in range(world)]
    # the same device twice, no devices, bad counts
    twice = (_int * 2)(1, 1)
    two = (_vp * 2)()
    assert lib.rr_comm_init_all(two, 2, twice) == -4
    assert b"twice" in lib.rr_last_error()
    assert lib.rr_comm_init_all(two, 0, None) == -2
    assert lib.rr_comm_init_all(None, 2, None) == -1
