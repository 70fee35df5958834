"""czk_net_* (include/czk.h): mpc-net's primitives and the reference's batch opens behind the C ABI.

CPU tests: the SHM transport on host buffers with one process per party (the control block, the generation barrier, chunked
staging slots, stats by the reference's counting rules, the timeout) and czk_sha256 against hashlib.
GPU tests: the batch opens on device lanes with 2 / 3 processes sharing the GPU (SHM) against the checker's arithmetic; the party
layout of a compiled C++ host against every other layout's digest; the RCCL transport with world = 1 (RCCL refuses two ranks on
one device) and with 2 ranks where two GPUs are visible; the wire format against the reference's Vec<Fr> serialisation."""
import hashlib
import os
import sys

import numpy as np
import pytest

from util import limbs_to_ints, rand_fr_canonical

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _spawn(target, world, *args, timeout=280):
    import multiprocessing as mp
    import queue
    import time
    c = mp.get_context("spawn")
    q = c.Queue()
    procs = [c.Process(target=target, args=(r, world, q) + args) for r in range(world)]
    for p in procs:
        p.start()
    res, t0 = [], time.time()
    while len(res) < world:
        try:
            res.append(q.get(timeout=1.0))
        except queue.Empty:
            dead = [(i, p.exitcode) for i, p in enumerate(procs) if p.exitcode not in (None, 0)]
            if dead or time.time() - t0 > timeout:            # a rank that died never reports: do not wait out the timeout for it
                for p in procs:
                    if p.is_alive():
                        p.kill()                                # exactly the processes this call started
                raise AssertionError(f"ranks died (rank, exit code): {dead}" if dead else f"ranks did not report within {timeout} s")
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(res)


def test_sha256_matches_hashlib():
    sys.path.insert(0, ROOT)
    from czk_amd import binding
    for n in (0, 1, 55, 56, 57, 63, 64, 65, 119, 120, 1000, 4097):
        data = bytes((i * 131 + n) & 255 for i in range(n))
        assert binding.sha256(data) == hashlib.sha256(data).digest(), n


def _payload(rank, nbytes):
    return ((np.arange(nbytes, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(977 * rank + 13)) >> np.uint64(5)).astype(np.uint8)


def _shm_host_worker(rank, world, q, idb, slot_bytes):
    sys.path.insert(0, ROOT)
    import czk_amd
    net = czk_amd.Net(None, czk_amd.CZK_NET_SHM, rank, world, idb, options={"slot_bytes": slot_bytes, "timeout_ms": 60000})
    ok = net.rank == rank and net.world == world
    sizes = (1, 63, 64, 65, 1000, 4096, 10007, 0)            # below, at and above the slot size; several chunk steps; the empty message
    for nbytes in sizes:
        got = net.broadcast(_payload(rank, nbytes))
        ok = ok and got.shape == (world, nbytes) and all(np.array_equal(got[p], _payload(p, nbytes)) for p in range(world))
        king = net.send_to_king(_payload(rank + 7, nbytes))
        ok = ok and ((king is None) if rank else all(np.array_equal(king[p], _payload(p + 7, nbytes)) for p in range(world)))
        parts = np.stack([_payload(100 + p, nbytes) for p in range(world)]) if rank == 0 else None
        mine = net.recv_from_king(parts, nbytes=nbytes)
        ok = ok and np.array_equal(mine, _payload(100 + rank, nbytes))
    net.barrier()
    st = net.stats()
    tot = sum(sizes)
    # mpc-net/src/multi.rs:148-150, 179-193, 214-220
    want = {"broadcasts": len(sizes), "to_king": len(sizes), "from_king": len(sizes),
            "bytes_sent": (world - 1) * tot + ((world - 1) * (tot + 8 * len(sizes)) if rank == 0 else tot),
            "bytes_recv": (world - 1) * tot + ((world - 1) * tot if rank == 0 else tot)}
    ok = ok and st == want
    net.stats_reset()
    ok = ok and sum(net.stats().values()) == 0
    # device buffers without a context are refused, not dereferenced
    try:
        net.broadcast(1234, nbytes=8, recv=5678, mem=czk_amd.CZK_MEM_DEVICE)
        ok = False
    except czk_amd.CzkError as e:
        ok = ok and e.code == 3
    net.close()
    q.put((rank, bool(ok)))


@pytest.mark.parametrize("world,slot_bytes", [(2, 64), (3, 4096), (4, 1 << 20), (8, 1 << 16)])
def test_shm_transport_host_buffers(world, slot_bytes):
    """mpc-net's three primitives between `world` processes through the shared-memory transport, no GPU involved"""
    res = _spawn(_shm_host_worker, world, os.urandom(16), slot_bytes)
    assert res == [(r, True) for r in range(world)]


def _ipc_dry_worker(rank, world, q, idb, slot_bytes):
    sys.path.insert(0, ROOT)
    import czk_amd
    ok = True
    try:
        czk_amd.Net(None, czk_amd.CZK_NET_IPC, rank, world, idb)       # the PRODUCT library has no stand-in: device mailboxes need a context
        ok = False
    except czk_amd.CzkError as e:
        ok = e.code == 3
    net = czk_amd.Net(None, czk_amd.CZK_NET_IPC, rank, world, idb, options={"slot_bytes": slot_bytes, "timeout_ms": 60000}, lab=True)
    for nbytes in (1, slot_bytes - 1, slot_bytes, slot_bytes + 1, 5 * slot_bytes + 3, 0):      # one and several chunk steps: the slots alternate by parity
        got = net.broadcast(_payload(rank, nbytes))
        ok = ok and got.shape == (world, nbytes) and all(np.array_equal(got[p], _payload(p, nbytes)) for p in range(world))
        king = net.send_to_king(_payload(rank + 7, nbytes))
        ok = ok and ((king is None) if rank else all(np.array_equal(king[p], _payload(p + 7, nbytes)) for p in range(world)))
        parts = np.stack([_payload(100 + p, nbytes) for p in range(world)]) if rank == 0 else None
        ok = ok and np.array_equal(net.recv_from_king(parts, nbytes=nbytes), _payload(100 + rank, nbytes))
    net.barrier()
    net.close()
    q.put((rank, bool(ok)))


@pytest.mark.parametrize("world,slot_bytes", [(2, 64), (3, 4096), (4, 192)])
def test_ipc_mailbox_hand_over_dry_run(world, slot_bytes):
    """The hipIpc transport's open order with every peer on ANOTHER device, on a box without GPUs (lab library, CZK_NET_IPC without a context): each rank
    allocates its mailbox, publishes a handle that names its rank and device (device = rank, so device != peer device for every pair), all ranks meet,
    each opens every peer's handle from the shared table, all meet again, and the three mpc-net primitives run through the mapped mailboxes by slot
    parity -- the code path of `czk-ipc` between GPUs with the three HIP calls (hipMalloc, hipIpcGetMemHandle, hipIpcOpenMemHandle with
    hipIpcMemLazyEnablePeerAccess) replaced by POSIX segments.  What is left untested until a multi-GPU lease is those three calls and the peer copies."""
    res = _spawn(_ipc_dry_worker, world, os.urandom(16), slot_bytes)
    assert res == [(r, True) for r in range(world)]


def _shm_timeout_worker(rank, world, q, idb):
    sys.path.insert(0, ROOT)
    import czk_amd
    net = czk_amd.Net(None, czk_amd.CZK_NET_SHM, rank, world, idb, options={"timeout_ms": 1500})
    code = 0
    if rank == 0:                                             # rank 1 never joins this exchange: rank 0 must come back with CZK_ERR_NET
        try:
            net.broadcast(_payload(0, 16))
        except czk_amd.CzkError as e:
            code = e.code
    else:
        import time
        time.sleep(3.0)
        try:                                                  # ... and a late peer finds the communicator aborted instead of hanging
            net.broadcast(_payload(1, 16))
        except czk_amd.CzkError as e:
            code = e.code
    net.close()
    q.put((rank, code))


def test_shm_transport_times_out_instead_of_hanging():
    assert _spawn(_shm_timeout_worker, 2, os.urandom(16), timeout=60) == [(0, 5), (1, 5)]


def test_net_create_rejects_bad_arguments():
    sys.path.insert(0, ROOT)
    import czk_amd
    for kw in (dict(rank=2, world=2), dict(rank=0, world=0), dict(rank=-1, world=1)):
        with pytest.raises(czk_amd.CzkError):
            czk_amd.Net(None, czk_amd.CZK_NET_SHM, kw["rank"], kw["world"], b"x")
    with pytest.raises(czk_amd.CzkError):
        czk_amd.Net(None, czk_amd.CZK_NET_SHM, 0, 1, b"y" * 33)          # SHM ids are 1..32 bytes
    with pytest.raises(czk_amd.CzkError):
        czk_amd.Net(None, czk_amd.CZK_NET_RCCL, 0, 1, b"z" * 128)        # RCCL needs a context
    net = czk_amd.Net(None, czk_amd.CZK_NET_SHM, 0, 1, os.urandom(8))    # a world of one: every primitive is a local copy
    assert np.array_equal(net.broadcast(_payload(3, 100))[0], _payload(3, 100))
    assert np.array_equal(net.recv_from_king(_payload(4, 50).reshape(1, -1)), _payload(4, 50))
    with pytest.raises(czk_amd.CzkError):
        net.set_option("exchange", 7)
    net.close()


# ---- GPU: the batch opens on device lanes --------------------------------------------------------------------------------------------
def _spdz_inputs(orc, world, n):
    """additive shares of `secret` and of alpha * secret; party p's MAC key share is alpha_p (share/spdz.rs:166-185 with a real key)"""
    secret = orc.fr_from_repr(rand_fr_canonical(1, n))
    alpha = orc.fr_from_repr(rand_fr_canonical(2, world))
    asum = alpha[0].reshape(1, 4)
    for p in range(1, world):
        asum = orc.fr_add(asum, alpha[p].reshape(1, 4))
    macv = orc.fr_mul(secret, np.tile(asum[0], (n, 1)))

    def shares_of(total, seed):
        parts = [orc.fr_from_repr(rand_fr_canonical(seed + p, n)) for p in range(world - 1)]
        rest = total
        for s in parts:
            rest = orc.fr_sub(rest, s)
        return parts + [rest]
    return secret, alpha, shares_of(secret, 10), shares_of(macv, 50)


def _open_worker(rank, world, q, idb, transport, share_device, n, slot_bytes):
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    # single node: RCCL's bootstrap on the loopback interface, no InfiniBand / MSCCL probing (they can stall for minutes on a box without network)
    for k, v in (("NCCL_SOCKET_IFNAME", "lo"), ("NCCL_IB_DISABLE", "1"), ("RCCL_MSCCL_ENABLE", "0"), ("RCCL_MSCCLPP_ENABLE", "0")):
        os.environ.setdefault(k, v)
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import torch
    import czk_amd
    import orc
    dev = 0 if share_device else rank
    torch.cuda.set_device(dev)
    ctx = czk_amd.Context(dev)
    net = czk_amd.Net(ctx, transport, rank, world, idb, options={"slot_bytes": slot_bytes} if transport != czk_amd.CZK_NET_RCCL else None)
    secret, alpha, shs, macs = _spdz_inputs(orc, world, n)
    sh = torch.from_numpy(shs[rank].view(np.int64)).cuda()
    mac = torch.from_numpy(macs[rank].view(np.int64)).cuda()
    out = torch.empty_like(sh)
    fails = []

    def check(tag, cond):
        if not cond:
            fails.append(tag)

    def host(t):
        ctx.sync()
        return t.cpu().numpy().view(np.uint64)
    for exch in (0, 1):                                      # ring and p2p deliver the same bytes
        net.set_option("exchange", exch)
        for commit in (False, True):
            out.zero_()
            torch.cuda.synchronize()                          # torch's stream and the context's private stream are not ordered by themselves
            check(1, net.spdz_batch_open(sh.data_ptr(), mac.data_ptr(), alpha[rank], n, out.data_ptr(), commit=commit) == 0)
            check(2, np.array_equal(host(out), secret))
        out.zero_()
        torch.cuda.synchronize()
        net.add_batch_open(sh.data_ptr(), n, out.data_ptr())
        check(3, np.array_equal(host(out), secret))
    # in place (out_value aliases the sh lane), as a caller that drops the share after opening it does
    tmp = sh.clone()
    torch.cuda.synchronize()
    check(4, net.spdz_batch_open(tmp.data_ptr(), mac.data_ptr(), alpha[rank], n, tmp.data_ptr()) == 0 and np.array_equal(host(tmp), secret))
    # a tampered MAC share trips the check on EVERY party, by the number of tampered elements
    bad_mac = mac.clone()
    if rank == world - 1:
        bad_mac[5, 0] ^= 1
        bad_mac[n - 1, 3] ^= 4
    torch.cuda.synchronize()
    check(5, net.spdz_batch_open(sh.data_ptr(), bad_mac.data_ptr(), alpha[rank], n, out.data_ptr()) == 2)
    # atomic_broadcast: the gathered vectors, deterministic commitment bytes; then a party whose data differs from what it committed to
    allx = torch.empty((world, n, 4), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    net.atomic_broadcast(sh.data_ptr(), n, allx.data_ptr(), rand32=bytes([rank + 1]) * 32)
    check(6, all(np.array_equal(host(allx)[p], shs[p]) for p in range(world)))
    # GSZ: degree-t shares; batch open, then king_compute with f = identity (the degree reduction of a product share)
    if world in (1, 2, 3, 4, 6, 8):
        t = (world - 1) // 2
        coeffs = orc.fr_from_repr(rand_fr_canonical(99, n * (t + 1))).reshape(n, t + 1, 4)
        mine = torch.from_numpy(np.stack([orc.gsz_share(coeffs[i], world)[rank] for i in range(n)]).view(np.int64)).cuda()
        check(7, net.gsz_batch_open(mine.data_ptr(), n, out.data_ptr(), degree=t) == 0 and np.array_equal(host(out), coeffs[:, 0]))
        if t >= 1:
            check(8, net.gsz_batch_open(mine.data_ptr(), n, out.data_ptr(), degree=t - 1) == n)
        bad = net.gsz_batch_king_compute(mine.data_ptr(), n, out.data_ptr(), degree=t)
        check(9, bad == 0 and np.array_equal(host(out), coeffs[:, 0]))     # every party's new share is the opened value (gsz20/mod.rs:508-512)
        if t >= 1:
            bad = net.gsz_batch_king_compute(mine.data_ptr(), n, out.data_ptr(), degree=t - 1)
            check(10, bad == (n if rank == 0 else 0))
    # king gather / scatter on Fr lanes
    g = torch.zeros((world, n, 4), dtype=torch.int64, device="cuda") if rank == 0 else None
    torch.cuda.synchronize()                                  # (the fill runs on torch's stream, the gather on the context's)
    net.fr_send_to_king(sh.data_ptr(), n, g.data_ptr() if g is not None else None)
    if rank == 0:
        check(11, all(np.array_equal(host(g)[p], shs[p]) for p in range(world)))
        parts = torch.from_numpy(np.stack(macs).view(np.int64)).cuda()
    torch.cuda.synchronize()
    net.fr_recv_from_king(parts.data_ptr() if rank == 0 else None, n, out.data_ptr())
    check(12, np.array_equal(host(out), macs[rank]))
    st = net.stats()
    check(13, st["broadcasts"] > 0 and st["to_king"] >= 1 and st["from_king"] >= 1)
    net.barrier()
    net.close()
    ctx.close()
    q.put((rank, not fails, fails))


@pytest.mark.gpu
@pytest.mark.parametrize("world,n,slot_bytes", [(2, 3000, 16 << 20), (3, 70001, 1 << 20)])
def test_batch_opens_processes_sharing_one_gpu(world, n, slot_bytes):
    """SpdzFieldShare / AdditiveFieldShare / GszFieldShare::batch_open, atomic_broadcast, king_compute and the king gather / scatter
    as single C-ABI calls on device lanes, one process per party on ONE GPU (SHM transport; 70001 x 32 B crosses the 1 MiB slots
    in three chunk steps), against the checker's field arithmetic"""
    import czk_amd
    res = _spawn(_open_worker, world, os.urandom(16), czk_amd.CZK_NET_SHM, True, n, slot_bytes)
    assert res == [(r, True, []) for r in range(world)]


@pytest.mark.gpu
@pytest.mark.parametrize("world,n,slot_bytes", [(2, 3000, 16 << 20), (3, 70001, 1 << 20)])
def test_batch_opens_over_device_mailboxes(world, n, slot_bytes):
    """the same opens through the IPC transport: the staging slots are device memory (one mailbox per rank, mapped into the peers with hipIpc),
    so with the parties on one GPU nothing of an exchange leaves HBM"""
    import czk_amd
    res = _spawn(_open_worker, world, os.urandom(16), czk_amd.CZK_NET_IPC, True, n, slot_bytes)
    assert res == [(r, True, []) for r in range(world)]


@pytest.mark.gpu
def test_rccl_transport_world_of_one():
    """The RCCL transport loads (dlopen librccl.so.1), initialises a communicator and runs every primitive and open with world = 1 --
    all a one-GPU box can execute of it (RCCL refuses two ranks on one device)."""
    import czk_amd
    idb = czk_amd.Net.unique_id(czk_amd.CZK_NET_RCCL)
    assert len(idb) == 128
    res = _spawn(_open_worker, 1, idb, czk_amd.CZK_NET_RCCL, True, 5000, 0)
    assert res == [(0, True, [])]


@pytest.mark.gpu
def test_rccl_transport_two_gpus():
    import torch
    import czk_amd
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    res = _spawn(_open_worker, 2, czk_amd.Net.unique_id(czk_amd.CZK_NET_RCCL), czk_amd.CZK_NET_RCCL, False, 50000, 0)
    assert res == [(0, True, []), (1, True, [])]


def _cheat_worker(rank, world, q, idb):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import torch
    import czk_amd
    torch.cuda.set_device(0)
    ctx = czk_amd.Context(0)
    net = czk_amd.Net(ctx, czk_amd.CZK_NET_SHM, rank, world, idb)
    n = 100
    x = torch.from_numpy(rand_fr_canonical(5 + rank, n).view(np.int64)).cuda()
    torch.cuda.synchronize()
    ctx.fr_from_repr(x.data_ptr(), out=x.data_ptr(), n=n, mem=1)
    ctx.sync()
    allx = torch.empty((world, n, 4), dtype=torch.int64, device="cuda")
    code = 0
    # party 1 commits to its vector and then sends another one: reproduce the two rounds by hand with the byte primitives
    import hashlib as hl
    wire = ctx.fr_vec_serialize(x.data_ptr(), n=n, mem=1)
    rnd = bytes([7 + rank]) * 32
    commit = np.frombuffer(hl.sha256(wire + rnd).digest(), dtype=np.uint8)
    if rank == 1:
        net.broadcast(commit)
        x[3, 0] ^= 1                                          # changes its mind after committing
        torch.cuda.synchronize()
        net.broadcast(x.data_ptr(), nbytes=32 * n, recv=allx.data_ptr(), mem=1)
        net.broadcast(np.frombuffer(rnd, dtype=np.uint8))
    else:
        try:
            net.atomic_broadcast(x.data_ptr(), n, allx.data_ptr(), rand32=rnd)
        except czk_amd.CzkError as e:
            code = e.code
    net.close()
    ctx.close()
    q.put((rank, code))


@pytest.mark.gpu
def test_atomic_broadcast_detects_a_vector_that_differs_from_its_commitment():
    """channel.rs:63-66: the receiver re-hashes what the other party sent and compares with the commitment of round 1"""
    assert _spawn(_cheat_worker, 2, os.urandom(16)) == [(0, 6), (1, 0)]


@pytest.mark.gpu
def test_wire_format_through_the_abi(orc):
    """czk_fr_vec_serialize / _deserialize = Vec<Fr>::serialize (u64 LE length + 32 LE bytes of into_repr per element,
    serialize/src/lib.rs:220-229), host and device memory; malformed input is refused like the reference's deserialize"""
    import torch
    import czk_amd
    ctx = czk_amd.Context(0)
    x = orc.fr_from_repr(rand_fr_canonical(3, 1000))
    want = (1000).to_bytes(8, "little") + b"".join(v.to_bytes(32, "little") for v in limbs_to_ints(orc.fr_into_repr(x)))
    assert ctx.fr_vec_serialize(x) == want
    xd = torch.from_numpy(x.view(np.int64)).cuda()
    assert ctx.fr_vec_serialize(xd.data_ptr(), n=1000, mem=1) == want
    assert np.array_equal(ctx.fr_vec_deserialize(want), x)
    back = torch.zeros((1000, 4), dtype=torch.int64, device="cuda")
    assert ctx.fr_vec_deserialize(want, out=back.data_ptr(), cap=1000, mem=1) == 1000
    ctx.sync()
    assert np.array_equal(back.cpu().numpy().view(np.uint64), x)
    assert ctx.fr_vec_serialize(np.zeros((0, 4), np.uint64)) == bytes(8)
    assert ctx.fr_vec_deserialize(bytes(8)).shape == (0, 4)
    for bad in (want[:-1], want + b"\0", (999).to_bytes(8, "little") + want[8:], want[:4]):
        with pytest.raises(czk_amd.CzkError):
            ctx.fr_vec_deserialize(bad)
    from util import R_MOD
    with pytest.raises(czk_amd.CzkError):                      # a value >= r is not a field element (from_repr -> None)
        ctx.fr_vec_deserialize((1).to_bytes(8, "little") + R_MOD.to_bytes(32, "little"))
    with pytest.raises(czk_amd.CzkError):
        ctx.fr_vec_deserialize(want, out=back.data_ptr(), cap=999, mem=1)
    ctx.close()


# ---- the party layout from a compiled host: tools/host_demo.cpp `party` ------------------------------------------------------------------
def _json_tail(proc):
    import json
    assert proc.returncode == 0, proc.stdout[-1500:] + __import__('util').child_errors(proc.stderr)
    return json.loads(proc.stdout.strip().splitlines()[-1])


@pytest.mark.gpu
@pytest.mark.parametrize("world,size,extra", [(2, ["--log-n", "12"], []), (3, ["--log-n", "10"], ["--no-commit-opens"]), (2, ["--constraints", "1000"], ["--no-tables", "--exchange", "p2p"]),
                                              (3, ["--log-n", "11"], ["--transport", "ipc"])])
def test_cpp_party_layout_yields_the_digest_of_every_other_layout(world, size, extra):
    """One process per MPC party, each a C++ host over include/czk.hpp whose opens run through czk::Net (SHM transport: the processes
    share this box's GPU): the proof's group elements (digest over all parties' affine results, bench.py's order) must equal
    (a) the same C++ host with all parties' lanes in one process, (b) bench.py's one-GPU layout, (c) bench.py's party layout over
    torch.distributed and (d) over czk_net -- five hosts / layouts, one digest."""
    import subprocess
    from test_abi import _build_host_demo
    exe = _build_host_demo()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    tables = [] if "--no-tables" not in extra else ["--no-tables"]

    cpp_party = _json_tail(__import__("util").run_ranks([exe, "party-launch", "--world", str(world), "--steps", "2", "--warmup", "1"] + size + extra,
                                          capture_output=True, text=True, timeout=600, env=env))
    assert cpp_party["layout"] == "party" and cpp_party["parties"] == world and cpp_party["share_lanes_per_process"] == 2
    # 2 opens per proof, each one broadcast of the sh lane + one (atomic: two) of dx_t -- the reference's message count (spdz.rs:166-185)
    per_open = 2 if "--no-commit-opens" in extra else 3   # commit-then-open is the default, as in the reference (spdz.rs:179)
    assert cpp_party["commit_opens"] == ("--no-commit-opens" not in extra)
    assert cpp_party["king_net_stats"]["broadcasts"] == 2 * 2 * per_open
    cpp_one = _json_tail(subprocess.run([exe, "bench", "--parties", str(world), "--steps", "2", "--warmup", "1"] + size + tables,
                                        capture_output=True, text=True, timeout=600, env=env))
    common = size + tables + ["--parties", str(world), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-seam-report", "--no-other-workloads"]
    py_one = _json_tail(subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT))
    py_party = _json_tail(__import__('util').run_ranks([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--layout", "party", "--backend", "gloo", "--device", "0"]
                                         + common, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT))
    # (d) the Python host in the party layout with the opens through the SAME communicator calls the C++ host makes (bench.py --net czk:
    # parallel.use_net; shared-memory transport, torch.distributed only carries the communicator id)
    py_party_czk = _json_tail(__import__('util').run_ranks([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--layout", "party", "--backend", "gloo", "--device", "0",
                                              "--net", "czk"] + common + (["--no-commit-opens"] if "--no-commit-opens" in extra else []),
                                             capture_output=True, text=True, timeout=600, env=env, cwd=ROOT))
    assert py_one["results_checked"] and py_party["results_checked"] and py_party_czk["results_checked"]
    assert py_party["net"] == "torch.distributed" and py_party_czk["net"] == "czk_net shm"
    all_digests = (cpp_party["results_sha256"], cpp_one["results_sha256"], py_one["config"]["results_sha256"], py_party["config"]["results_sha256"],
                   py_party_czk["config"]["results_sha256"])
    assert len(set(all_digests)) == 1, all_digests


@pytest.mark.gpu
@pytest.mark.parametrize("transport", ["ipc", "shm"])
def test_cpp_party_layout_with_eight_parties(transport):
    """BASELINE configs[4]'s party count (8 parties, mpc-net/src/multi.rs:15-23 with eight hosts): eight C++ party processes sharing this
    box's GPU, opens through czk::Net, against the same host with all sixteen share lanes in one process -- one digest."""
    import subprocess
    from test_abi import _build_host_demo
    exe = _build_host_demo()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    size = ["--log-n", "10", "--steps", "2", "--warmup", "1"]
    party = _json_tail(__import__("util").run_ranks([exe, "party-launch", "--world", "8", "--transport", transport] + size,
                                                    capture_output=True, text=True, timeout=600, env=env))
    assert party["parties"] == 8 and party["share_lanes_per_process"] == 2 and party["transport"] == transport
    assert party["commit_opens"] is True and party["king_net_stats"]["broadcasts"] == 2 * 2 * 3   # 2 proofs x 2 opens x (sh lanes + dx_t through atomic_broadcast's two rounds)
    one = _json_tail(subprocess.run([exe, "bench", "--parties", "8"] + size, capture_output=True, text=True, timeout=600, env=env))
    assert one["share_lanes"] == 16 and one["mac_check_failures"] == 0
    assert party["results_sha256"] == one["results_sha256"]
