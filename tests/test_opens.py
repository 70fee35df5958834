"""The `open` protocols on device buffers (scope rows f1 / f4): SPDZ batch_open round by round and GSZ / Shamir batch_open,
against the checker's restatements (oracle/: share/spdz.rs:166-185, share/gsz20/mod.rs:286-300, 434-466), single process with
the parties' contributions stacked as lanes, and as real multi-rank runs (gloo on the 1-GPU box; RCCL when >= 2 GPUs)."""
import os
import socket
import sys

import numpy as np
import pytest

from util import ints_to_limbs, rand_fr_canonical

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ctx():
    import czk_amd
    c = czk_amd.Context(0)
    yield c
    c.close()


@pytest.mark.gpu
def test_spdz_open_round_by_round_matches_reference_arithmetic(ctx, orc):
    import torch
    parties, n = 3, 5000
    secret = orc.fr_from_repr(rand_fr_canonical(1, n))
    alpha = orc.fr_from_repr(rand_fr_canonical(2, parties))                     # MAC key shares (mac_share of each party)
    alpha_sum = alpha[0]
    for p in range(1, parties):
        alpha_sum = orc.fr_add(alpha_sum.reshape(1, 4), alpha[p].reshape(1, 4))[0]
    sh = [orc.fr_from_repr(rand_fr_canonical(10 + p, n)) for p in range(parties - 1)]
    rest = secret
    for s in sh:
        rest = orc.fr_sub(rest, s)
    sh.append(rest)
    mac_total = orc.fr_mul(secret, np.tile(alpha_sum, (n, 1)))                  # alpha * x, additively shared
    mac = [orc.fr_from_repr(rand_fr_canonical(20 + p, n)) for p in range(parties - 1)]
    rest = mac_total
    for m in mac:
        rest = orc.fr_sub(rest, m)
    mac.append(rest)
    shd = torch.from_numpy(np.stack(sh).view(np.int64)).cuda()
    vals = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    assert ctx.fr_lanes_sum(shd.data_ptr(), parties, n, out_ptr=vals.data_ptr()) == 0
    ctx.sync()
    assert np.array_equal(vals.cpu().numpy().view(np.uint64), secret)
    dxs = []
    for p in range(parties):
        macd = torch.from_numpy(mac[p].view(np.int64)).cuda()
        dx = torch.empty((n, 4), dtype=torch.int64, device="cuda")
        ctx.fr_spdz_dx(vals.data_ptr(), macd.data_ptr(), alpha[p], dx.data_ptr(), n)
        ctx.sync()
        want = orc.fr_sub(orc.fr_mul(np.tile(alpha[p], (n, 1)), secret), mac[p])  # mac_share * val - mac (spdz.rs:176-180)
        assert np.array_equal(dx.cpu().numpy().view(np.uint64), want)
        dxs.append(dx)
    alld = torch.stack(dxs).contiguous()
    assert ctx.fr_lanes_sum(alld.data_ptr(), parties, n, count_nonzero=True) == 0
    alld[1, 77, 0] ^= 1                                                          # a cheating party
    alld[2, 4000, 3] ^= 5
    assert ctx.fr_lanes_sum(alld.data_ptr(), parties, n, count_nonzero=True) == 2


@pytest.mark.gpu
@pytest.mark.parametrize("parties,deg", [(1, 0), (2, 0), (3, 1), (4, 1), (6, 2), (8, 3), (12, 5), (24, 11)])
def test_gsz_open_matches_reference_restatement(ctx, orc, parties, deg):
    import torch
    n = 1500
    coeffs = orc.fr_from_repr(rand_fr_canonical(parties, n * (deg + 1))).reshape(n, deg + 1, 4)
    shares = np.stack([orc.gsz_share(coeffs[i], parties) for i in range(n)], axis=1)       # (parties, n, 4): party j holds p_i(w^j)
    k = ctx.share_domain_constants(parties)
    assert np.array_equal(k["group_gen"], orc.fr_root_of_unity_mixed(parties))
    sd = torch.from_numpy(shares.view(np.int64)).cuda()
    out = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    assert ctx.fr_gsz_open(sd.data_ptr(), parties, n, out.data_ptr(), degree=deg) == 0
    got = out.cpu().numpy().view(np.uint64)
    want, bad = orc.gsz_open(shares, degree=deg)
    assert bad == 0 and np.array_equal(got, want) and np.array_equal(got, coeffs[:, 0])    # the secret is p(0)
    if deg >= 1:
        # a tighter bound must flag every element (top coefficient is non-zero with overwhelming probability) -- like the checker
        assert ctx.fr_gsz_open(sd.data_ptr(), parties, n, out.data_ptr(), degree=deg - 1) == orc.gsz_open(shares, degree=deg - 1)[1] == n
        # per-element bounds; a corrupted share breaks the bound only where there is slack to detect it
        degs = np.full(n, deg, dtype=np.uint32)
        degs[::7] = deg - 1
        dd = torch.from_numpy(degs.view(np.int32)).cuda()
        sd2 = sd.clone()
        sd2[parties - 1, 3, 0] ^= 1
        bad_gpu = ctx.fr_gsz_open(sd2.data_ptr(), parties, n, out.data_ptr(), degrees_ptr=dd.data_ptr())
        sh2 = sd2.cpu().numpy().view(np.uint64)
        want2, bad_ref = orc.gsz_open(sh2, degrees=degs)
        assert bad_gpu == bad_ref and np.array_equal(out.cpu().numpy().view(np.uint64), want2)


@pytest.mark.gpu
def test_share_domain_rejects_party_counts_without_a_subgroup(ctx, orc):
    import czk_amd
    for parties in (5, 7, 9, 10, 18):
        assert orc.fr_root_of_unity_mixed(parties) is None
        with pytest.raises(czk_amd.CzkError) as e:
            ctx.share_domain_constants(parties)
        assert e.value.code == 1


def test_wire_format_is_the_references_vec_serialization(orc):
    """Vec<Fr>::serialize = u64 LE length + 32 LE bytes of into_repr per element (serialize/src/lib.rs:220-229)."""
    sys.path.insert(0, ROOT)
    from czk_amd import parallel
    from util import limbs_to_ints
    x = orc.fr_from_repr(rand_fr_canonical(3, 9))
    rep = orc.fr_into_repr(x)
    buf = parallel.serialize_fr_vec(rep)
    want = (9).to_bytes(8, "little") + b"".join(v.to_bytes(32, "little") for v in limbs_to_ints(rep))
    assert buf == want
    assert np.array_equal(parallel.deserialize_fr_vec(buf), rep)
    with pytest.raises(ValueError):
        parallel.deserialize_fr_vec(buf[:-1])
    assert parallel.serialize_fr_vec(np.zeros((0, 4), np.uint64)) == bytes(8)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _king_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch
    from czk_amd import parallel
    parallel.init("gloo")
    x = torch.arange(6, dtype=torch.int64).reshape(3, 2) + 100 * rank
    got = parallel.send_to_king(x)
    ok = (got is None) if rank else (got.shape == (world, 3, 2) and all(bool((got[p] == torch.arange(6).reshape(3, 2) + 100 * p).all()) for p in range(world)))
    # king_compute: the king adds everything up and hands every party the total plus its rank
    back = parallel.king_compute(x, lambda xs: torch.stack([xs.sum(0) + p for p in range(world)]))
    total = sum(torch.arange(6, dtype=torch.int64).reshape(3, 2) + 100 * p for p in range(world))
    ok = ok and bool((back == total + rank).all())
    parallel.barrier()
    q.put((rank, ok))
    torch.distributed.destroy_process_group()


def test_king_gather_scatter_three_ranks_gloo():
    import torch.multiprocessing as mp
    world, port = 3, _free_port()
    c = mp.get_context("spawn")
    q = c.Queue()
    procs = [c.Process(target=_king_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(r, True) for r in range(world)]


def _open_worker(rank, world, port, q, backend, share_device):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import torch
    import czk_amd
    from czk_amd import parallel
    import orc
    dev = 0 if share_device else rank
    torch.cuda.set_device(dev)
    if backend == "nccl":
        torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", dev))
    else:
        parallel.init("gloo")
    ctx = czk_amd.Context(dev)
    n = 3000
    # SPDZ: additive shares of `secret` and of alpha * secret; party p's MAC key share is alpha_p
    secret = orc.fr_from_repr(rand_fr_canonical(1, n))
    alpha = orc.fr_from_repr(rand_fr_canonical(2, world))
    asum = alpha[0].reshape(1, 4)
    for p in range(1, world):
        asum = orc.fr_add(asum, alpha[p].reshape(1, 4))
    macv = orc.fr_mul(secret, np.tile(asum[0], (n, 1)))

    def share_of(total, seed):
        parts = [orc.fr_from_repr(rand_fr_canonical(seed + p, n)) for p in range(world - 1)]
        rest = total
        for s in parts:
            rest = orc.fr_sub(rest, s)
        return (parts + [rest])[rank]
    sh = torch.from_numpy(share_of(secret, 10).view(np.int64)).cuda()
    mac = torch.from_numpy(share_of(macv, 50).view(np.int64)).cuda()
    ok = True
    for method in ("ring", "p2p"):           # the two transports of the share exchange deliver the same bytes
        parallel.set_exchange(method)
        for commit in (True, False):
            vals = parallel.spdz_batch_open(ctx, sh, mac, alpha[rank], commit=commit)
            ctx.sync()
            ok = ok and np.array_equal(vals.cpu().numpy().view(np.uint64), secret)
        # the reference's honest-but-curious additive open (share/add.rs:256-259): the sum of the shares
        vals = parallel.additive_batch_open(ctx, sh)
        ctx.sync()
        ok = ok and np.array_equal(vals.cpu().numpy().view(np.uint64), secret)
        g1, g2 = parallel.all_gather_shares(sh, ctx, method="ring"), parallel.all_gather_shares(sh, ctx, method="p2p")
        ctx.sync()
        ok = ok and bool(torch.equal(g1, g2))
    parallel.set_exchange("ring")
    # a wrong MAC share must trip the check on every party
    bad_mac = mac.clone()
    if rank == world - 1:
        bad_mac[5, 0] ^= 1
    try:
        parallel.spdz_batch_open(ctx, sh, bad_mac, alpha[rank], commit=False)
        ok = False
    except parallel.MpcCheckError:
        pass
    # GSZ: degree-t shares of a vector of secrets
    if world in (2, 3, 4, 6, 8):
        t = (world - 1) // 2
        coeffs = orc.fr_from_repr(rand_fr_canonical(99, n * (t + 1))).reshape(n, t + 1, 4)
        mine = np.stack([orc.gsz_share(coeffs[i], world)[rank] for i in range(n)])
        out = parallel.gsz_batch_open(ctx, torch.from_numpy(mine.view(np.int64)).cuda(), t)
        ctx.sync()
        ok = ok and np.array_equal(out.cpu().numpy().view(np.uint64), coeffs[:, 0])
    parallel.barrier(torch.cuda.synchronize)
    q.put((rank, bool(ok)))
    ctx.close()
    torch.distributed.destroy_process_group()


def _run_open_ranks(world, backend, share_device):
    import torch.multiprocessing as mp
    port = _free_port()
    c = mp.get_context("spawn")
    q = c.Queue()
    procs = [c.Process(target=_open_worker, args=(r, world, port, q, backend, share_device)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=280) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(r, True) for r in range(world)]


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_opens_one_process_per_party_shared_gpu(world):
    """spdz_batch_open (two rounds, with and without the commitment) and gsz_batch_open with one process per party; the ranks
    share this box's single GPU, so the exchange runs over gloo."""
    _run_open_ranks(world, "gloo", True)


@pytest.mark.gpu
def test_opens_over_rccl_when_two_gpus_are_present():
    """The RCCL branch of the exchange (all_gather_into_tensor, gather / scatter): runs wherever >= 2 GPUs are visible."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    _run_open_ranks(2, "nccl", False)


def _exchange_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch
    from czk_amd import parallel
    parallel.init("gloo")
    torch.manual_seed(1234 + rank)
    x = torch.randint(-2**62, 2**62, (257, 4), dtype=torch.int64)
    ring = parallel.all_gather_shares(x, method="ring")
    p2p = parallel.all_gather_shares(x, method="p2p")
    ok = ring.shape == (world, 257, 4) and bool(torch.equal(ring, p2p)) and bool(torch.equal(ring[rank], x))
    parallel.set_exchange("p2p")
    ok = ok and parallel.get_exchange() == "p2p" and bool(torch.equal(parallel.all_gather_shares(x), ring))
    try:
        parallel.set_exchange("tree")
        ok = False
    except ValueError:
        pass
    parallel.barrier()
    q.put((rank, bool(ok)))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_p2p_exchange_equals_ring_all_gather_gloo(world):
    """the opens' share exchange as world - 1 grouped point-to-point copies (batch_isend_irecv; on an MI355X node every
    pair of GPUs has its own xGMI link) beside the ring all-gather, selectable -- identical bytes on every rank.  CPU, gloo."""
    import torch.multiprocessing as mp
    port = _free_port()
    c = mp.get_context("spawn")
    q = c.Queue()
    procs = [c.Process(target=_exchange_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(r, True) for r in range(world)]
