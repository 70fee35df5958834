"""One process per GPU.  The hot path shards by independent units (MPC parties / proofs: SURVEY.md section 8e), so
the data path needs no collective; torch.distributed (RCCL on GPUs, gloo on CPU) carries only (a) the timing
reduction of bench.py and (b) the mpc-net style share exchange of an `open` (mpc-net/src/multi.rs:145-173 is an
all-gather of one equal-length buffer per party; mpc-algebra/src/channel.rs:13-21)."""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend: str | None = None):
    """Initialises the default process group from the torchrun environment (no-op for a single process)."""
    rank, world, local_rank = env_rank_world()
    if world > 1 and not dist.is_initialized():
        if os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost", "::1"):
            # one node: sockets on the loopback interface (gloo / RCCL bootstrap otherwise resolve the host's name first -- minutes where the resolver times out)
            for k, v in (("GLOO_SOCKET_IFNAME", "lo"), ("NCCL_SOCKET_IFNAME", "lo"), ("NCCL_IB_DISABLE", "1")):
                os.environ.setdefault(k, v)
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, **kw)
    return rank, world, local_rank


class MpcCheckError(RuntimeError):
    """A protocol check of an open failed (SPDZ MAC check, GSZ degree bound, commitment mismatch): the reference `assert!`s these
    in release builds too (share/spdz.rs:181-184, share/gsz20/mod.rs:452), so they are exceptions here, never Python `assert`s."""


# How the share exchange of an open crosses the GPUs (RCCL backends; gloo always stages through the host):
#   "ring" : one all_gather_into_tensor -- RCCL's ring, every hop bound by one xGMI link
#   "p2p"  : world - 1 sends + world - 1 receives in ONE grouped batch_isend_irecv -- every pair of GPUs has its own xGMI link on an MI355X
#            node, so the n - 1 copies of a party's buffer travel concurrently over n - 1 links (mpc-net's own shape: a star of point-to-point
#            connections, mpc-net/src/multi.rs:145-173) instead of n - 1 ring steps.
# Both deliver identical bytes (tests/test_opens.py); which is faster is for the first multi-GPU lease to A/B.
_EXCHANGE = "ring"


def set_exchange(method: str):
    global _EXCHANGE
    if method not in ("ring", "p2p"):
        raise ValueError("exchange method must be 'ring' or 'p2p'")
    _EXCHANGE = method
    if _NET is not None:
        _NET.set_option("exchange", 1 if method == "p2p" else 0)


def get_exchange() -> str:
    return _EXCHANGE


# The opens' transport.  None: torch.distributed (RCCL through torch on GPUs, gloo through the host otherwise) -- the exchange is issued from
# Python on torch's stream.  A binding.Net (czk_net, include/czk.h): the SAME calls a compiled host makes -- RCCL / shared memory inside the
# library, enqueued on the czk context's stream; every function below then is a thin wrapper over one C-ABI call (use_net).
_NET = None


def use_net(net):
    """Route the opens through a czk_net communicator (binding.Net) instead of torch.distributed; None switches back."""
    global _NET
    _NET = net
    if net is not None:
        net.set_option("exchange", 1 if _EXCHANGE == "p2p" else 0)


def get_net():
    return _NET


def make_net(ctx, transport: str | None = None, device=None):
    """A czk_net over the ranks of torch's default process group (which only carries the communicator id here): transport "rccl" (one GPU
    per rank; default when the group's backend is nccl), "shm" (ranks of one node in any assignment to GPUs, staged through shared host memory) or "ipc" (the
    same with device mailboxes mapped between the processes: nothing leaves device memory)."""
    from . import binding
    world, rank = _world_rank()
    if transport is None:
        transport = "rccl" if (dist.is_initialized() and dist.get_backend() == "nccl") else "shm"
    t = {"rccl": binding.CZK_NET_RCCL, "shm": binding.CZK_NET_SHM, "ipc": binding.CZK_NET_IPC}[transport]
    box = [binding.Net.unique_id(t) if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(box, src=0, **({"device": device} if device is not None else {}))
    return binding.Net(ctx, t, rank, world, box[0])


def _net_in(ctx, t: torch.Tensor):
    """`t` may come from torch kernels on torch's current stream; czk_net enqueues on the context's stream"""
    if ctx.stream_handle() != torch.cuda.current_stream(t.device).cuda_stream:
        _ctx_stream(ctx, t.device).wait_stream(torch.cuda.current_stream(t.device))


def _ctx_stream(ctx, device):
    """torch view of the czk context's stream (czk_ctx_stream): lets torch's stream and the context's stream wait on each other with
    EVENTS instead of host synchronisation"""
    return torch.cuda.ExternalStream(ctx.stream_handle(), device=device)


def _before_exchange(ctx, t: torch.Tensor):
    """`t` was produced by the context's kernels; the exchange runs on torch's current stream (or through the host)."""
    if t.is_cuda and ctx is not None:
        torch.cuda.current_stream(t.device).wait_stream(_ctx_stream(ctx, t.device))     # event wait, no host block
    elif ctx is not None:
        ctx.sync()


def _settle(t: torch.Tensor, ctx=None):
    """After a collective on a CUDA backend: RCCL enqueues on torch's / its own stream and returns; the library's kernels that
    read the result run on the czk context's stream, which may be a private non-blocking one.  With the context at hand its stream
    waits on an event recorded behind the collective (no host synchronisation: the host goes on enqueueing); without it the host waits."""
    if t is not None and t.is_cuda:
        if ctx is not None:
            _ctx_stream(ctx, t.device).wait_stream(torch.cuda.current_stream(t.device))
        else:
            torch.cuda.current_stream(t.device).synchronize()
    return t


def _after_consume(ctx, t: torch.Tensor):
    """The context's kernels have been ENQUEUED on buffers torch allocated (the gathered shares, the result): order torch's current stream
    behind them, so that (a) torch work on the result and (b) torch's caching allocator re-using a buffer freed by the caller are both ordered
    after those kernels -- again an event wait, not a host synchronisation.  (No-op when the context shares torch's stream.)"""
    if t is not None and t.is_cuda and ctx is not None:
        torch.cuda.current_stream(t.device).wait_stream(_ctx_stream(ctx, t.device))
    return t


def partition_units(n_units: int, world: int, rank: int) -> list[int]:
    """Contiguous block partition of independent units (party-proofs) over ranks; sizes differ by at most one."""
    base, extra = divmod(n_units, world)
    start = rank * base + min(rank, extra)
    return list(range(start, start + base + (1 if rank < extra else 0)))


def barrier(device_sync=None):
    if device_sync:
        device_sync()
    if dist.is_initialized():
        dist.barrier()
    if device_sync:
        device_sync()


def max_over_ranks(seconds: float, device=None) -> float:
    if not dist.is_initialized():
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device=None) -> float:
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def aggregate_throughput(units_local: float, seconds_local: float, device=None) -> tuple[float, float]:
    """(whole-job units/s, max-over-ranks seconds): all ranks' units divided by the slowest rank's time."""
    t = max_over_ranks(seconds_local, device)
    return sum_over_ranks(units_local, device) / t, t


def _p2p_exchange(out: torch.Tensor, mine: torch.Tensor, world: int, rank: int):
    """out[p] <- party p's buffer: world - 1 isend + world - 1 irecv issued as one group"""
    out[rank].copy_(mine)
    ops = []
    for d in range(1, world):
        ops.append(dist.P2POp(dist.isend, mine, (rank + d) % world))
        ops.append(dist.P2POp(dist.irecv, out[(rank - d) % world], (rank - d) % world))
    for w in dist.batch_isend_irecv(ops):
        w.wait()      # CUDA backends: orders torch's current stream behind the transfer (no host block); gloo: completes it


def all_gather_shares(share: torch.Tensor, ctx=None, method: str | None = None) -> torch.Tensor:
    """mpc-net `broadcast`: every party contributes one equal-length buffer and receives all of them
    (mpc-net/src/multi.rs:145-173).  Returns a (world, *share.shape) tensor; the modular sum of an `open`
    (share/spdz.rs:166-185) is then a local, share-linear pointwise step.  `ctx`: the czk context whose kernels produced `share` and
    will consume the result -- given, the hand-over in both directions is an event wait between streams; omitted, the caller has
    synchronised and the host waits for the collective.  `method`: "ring" / "p2p" (default: set_exchange)."""
    if not dist.is_initialized() and _NET is None:
        return share.unsqueeze(0)
    if _NET is not None and share.is_cuda:
        c = ctx if ctx is not None else _NET.ctx
        share = share.contiguous()
        out = torch.empty((_NET.world,) + tuple(share.shape), dtype=share.dtype, device=share.device)
        _net_in(c, share)
        if method is not None:
            _NET.set_option("exchange", 1 if method == "p2p" else 0)
        _NET.broadcast(share.data_ptr(), nbytes=share.numel() * share.element_size(), recv=out.data_ptr(), mem=1)
        if method is not None:
            _NET.set_option("exchange", 1 if _EXCHANGE == "p2p" else 0)
        if ctx is None:
            c.sync()
        return _after_consume(c, out)
    world, rank = dist.get_world_size(), dist.get_rank()
    method = method or _EXCHANGE
    if ctx is not None:
        _before_exchange(ctx, share)
    if dist.get_backend() == "gloo":
        # CPU tests, and rigs without one GPU per party (several ranks on one device): stage through the host
        host = share.contiguous().cpu()
        if method == "p2p":
            stacked = torch.empty((world,) + tuple(host.shape), dtype=host.dtype)
            _p2p_exchange(stacked, host, world, rank)
            return _settle(stacked.to(share.device), ctx)
        out = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(out, host)
        return _settle(torch.stack(out).to(share.device), ctx)
    out = torch.empty((world,) + tuple(share.shape), dtype=share.dtype, device=share.device)
    if method == "p2p":
        _p2p_exchange(out, share.contiguous(), world, rank)
    else:
        dist.all_gather_into_tensor(out, share.contiguous())   # RCCL: one all-gather straight into the (world, ...) result
    return _settle(out, ctx)


# ------------------------------------------------------------------------------------------------------------------
# mpc-net's other primitives over torch.distributed (RCCL on GPUs, gloo on CPU) and the reference's wire format
# ------------------------------------------------------------------------------------------------------------------
def _world_rank():
    if _NET is not None:
        return _NET.world, _NET.rank
    return (dist.get_world_size(), dist.get_rank()) if dist.is_initialized() else (1, 0)


def send_to_king(x: torch.Tensor):
    """mpc-net `send_to_king` (mpc-net/src/multi.rs:175-210): the king (rank 0) receives every party's buffer as a
    (world, *x.shape) tensor in party order, everyone else None.  One grouped gather (RCCL: send/recv group)."""
    world, rank = _world_rank()
    if world == 1:
        return x.unsqueeze(0)
    if _NET is not None and x.is_cuda:
        x = x.contiguous()
        out = torch.empty((world,) + tuple(x.shape), dtype=x.dtype, device=x.device) if rank == 0 else None
        _net_in(_NET.ctx, x)
        _NET.send_to_king(x.data_ptr(), nbytes=x.numel() * x.element_size(), recv=out.data_ptr() if out is not None else None, mem=1)
        _after_consume(_NET.ctx, x)
        return out
    if dist.get_backend() == "gloo":
        host = x.contiguous().cpu()
        out = [torch.empty_like(host) for _ in range(world)] if rank == 0 else None
        dist.gather(host, out, dst=0)
        return torch.stack(out).to(x.device) if rank == 0 else None
    out = [torch.empty_like(x) for _ in range(world)] if rank == 0 else None
    dist.gather(x.contiguous(), out, dst=0)
    _settle(x)
    return torch.stack(out) if rank == 0 else None


def recv_from_king(xs, like: torch.Tensor) -> torch.Tensor:
    """mpc-net `recv_from_king` (multi.rs:211-242): the king hands party p the p-th of equally long buffers (xs: (world,
    *like.shape) on the king, None elsewhere); returns this party's buffer.  One grouped scatter."""
    world, rank = _world_rank()
    if world == 1:
        return xs[0]
    if _NET is not None and like.is_cuda:
        out = torch.empty_like(like, memory_format=torch.contiguous_format)
        if rank == 0:
            assert xs.shape[0] == world
            xs = xs.contiguous()
            _net_in(_NET.ctx, xs)
        _NET.recv_from_king(xs.data_ptr() if rank == 0 else None, nbytes=out.numel() * out.element_size(), recv=out.data_ptr(), mem=1)
        return _after_consume(_NET.ctx, out)
    gloo = dist.get_backend() == "gloo"
    out = torch.empty_like(like.cpu() if gloo else like)
    parts = None
    if rank == 0:
        assert xs.shape[0] == world
        parts = [(xs[p].contiguous().cpu() if gloo else xs[p].contiguous()) for p in range(world)]
    dist.scatter(out, parts, src=0)
    return _settle(out.to(like.device))


def king_compute(x: torch.Tensor, f):
    """mpc-algebra/src/channel.rs:77-80: send to the king, the king applies f to the stacked inputs, everyone gets its part back."""
    got = send_to_king(x)
    return recv_from_king(f(got) if got is not None else None, x)


COMMIT_RAND_BYTES = 32   # mpc-algebra/src/channel.rs:89


def serialize_fr_vec(canonical_limbs) -> bytes:
    """`Vec<Fr>::serialize` (algebra/serialize/src/lib.rs:220-229; fields/macros.rs impl_prime_field_serializer with EmptyFlags):
    u64 little-endian length, then 32 little-endian bytes of into_repr() per element.  Input: (n, 4) uint64 canonical limbs
    (czk_fr_into_repr's output) as a numpy array or CPU tensor."""
    import numpy as np
    a = canonical_limbs.numpy() if isinstance(canonical_limbs, torch.Tensor) else np.asarray(canonical_limbs)
    a = np.ascontiguousarray(a).view(np.uint64).reshape(-1, 4)
    return int(a.shape[0]).to_bytes(8, "little") + a.astype("<u8").tobytes()


def deserialize_fr_vec(buf: bytes):
    """Inverse of serialize_fr_vec: (n, 4) uint64 canonical limbs; raises on a length that does not match the payload."""
    import numpy as np
    n = int.from_bytes(buf[:8], "little")
    if len(buf) != 8 + 32 * n:
        raise ValueError("Vec<Fr> wire format: length prefix does not match the payload")
    return np.frombuffer(buf[8:], dtype="<u8").reshape(n, 4).astype(np.uint64)


def atomic_broadcast(ctx, x: torch.Tensor, rand32: bytes | None = None) -> torch.Tensor:
    """mpc-algebra/src/channel.rs:50-75 `atomic_broadcast` of one Fr vector (device tensor, Montgomery limbs): every party first
    broadcasts SHA-256(serialized vector || 32 random bytes), then the data itself; each receiver re-hashes what the others
    sent and compares (commit-then-open: nobody can choose its vector after seeing the others').  Returns the gathered
    vectors (world, n, 4).  The commitment runs over the reference's wire bytes (canonical limbs with a u64 length prefix), so
    the hashes equal the reference's for equal data and randomness.  Host-side hashing: this is the protocol's check, not hot-path
    arithmetic; pass commit=False to spdz_batch_open to skip it."""
    import hashlib
    import os as _os
    import numpy as np
    world, rank = _world_rank()
    n = x.shape[0]
    if _NET is not None and x.is_cuda:
        x = x.contiguous()
        data = torch.empty((world,) + tuple(x.shape), dtype=x.dtype, device=x.device)
        _net_in(ctx, x)
        try:
            _NET.atomic_broadcast(x.data_ptr(), n, data.data_ptr(), rand32=rand32)
        except Exception as e:
            if getattr(e, "code", None) == 6:            # CZK_ERR_CHECK
                raise MpcCheckError(str(e)) from e
            raise
        return _after_consume(ctx, data)
    rep = torch.empty_like(x)
    ctx.fr_into_repr(x.data_ptr(), out=rep.data_ptr(), n=n, mem=1)
    ctx.sync()
    wire = serialize_fr_vec(rep.cpu().numpy().view(np.uint64))
    rnd = rand32 if rand32 is not None else _os.urandom(COMMIT_RAND_BYTES)
    digest = hashlib.sha256(wire + rnd).digest()
    dev = x.device
    commits = all_gather_shares(torch.frombuffer(bytearray(digest), dtype=torch.uint8).to(dev))           # round 1: commitments
    data = all_gather_shares(x, ctx)                                                                         # round 2: the vectors ...
    rnds = all_gather_shares(torch.frombuffer(bytearray(rnd), dtype=torch.uint8).to(dev))                   # ... and the randomness
    for p in range(world):
        if p == rank:
            continue
        other = torch.empty_like(x)
        ctx.fr_into_repr(data[p].contiguous().data_ptr(), out=other.data_ptr(), n=n, mem=1)
        ctx.sync()
        w = serialize_fr_vec(other.cpu().numpy().view(np.uint64))
        if hashlib.sha256(w + bytes(rnds[p].cpu().numpy())).digest() != bytes(commits[p].cpu().numpy()):
            raise MpcCheckError(f"atomic_broadcast: party {p}'s data does not match its commitment")
    return data


def spdz_batch_open(ctx, sh: torch.Tensor, mac: torch.Tensor, mac_share, commit: bool = True) -> torch.Tensor:
    """SpdzFieldShare::batch_open as the reference runs it with one process per party (mpc-algebra/src/share/spdz.rs:166-185):
    round 1 broadcast of the `sh` lanes and their sum; dx_t = mac_share * value - mac; round 2 atomic_broadcast of dx_t, whose
    sum must vanish.  MAC shares themselves never leave the party.  sh, mac: (n, 4) device tensors; mac_share: (4,) uint64."""
    world, _ = _world_rank()
    n = sh.shape[0]
    if _NET is not None and sh.is_cuda:                                     # one C-ABI call: czk_spdz_batch_open
        sh, mac = sh.contiguous(), mac.contiguous()
        vals = torch.empty_like(sh)
        _net_in(ctx, sh)
        try:
            bad = _NET.spdz_batch_open(sh.data_ptr(), mac.data_ptr(), mac_share, n, vals.data_ptr(), commit=commit)
        except Exception as e:
            if getattr(e, "code", None) == 6:
                raise MpcCheckError(str(e)) from e
            raise
        if bad != 0:
            raise MpcCheckError(f"SPDZ MAC check failed on {bad} of {n} opened values")
        return _after_consume(ctx, vals)
    gathered = all_gather_shares(sh, ctx)                                  # Net::broadcast(&s_vals); stream hand-over by events (sh may be in flight)
    vals = torch.empty_like(sh)
    ctx.fr_lanes_sum(gathered.data_ptr(), world, n, out_ptr=vals.data_ptr())
    dx = torch.empty_like(sh)
    ctx.fr_spdz_dx(vals.data_ptr(), mac.data_ptr(), mac_share, dx.data_ptr(), n)
    all_dx = atomic_broadcast(ctx, dx) if commit else all_gather_shares(dx, ctx)   # Net::atomic_broadcast(&dx_ts)
    bad = ctx.fr_lanes_sum(all_dx.contiguous().data_ptr(), world, n, count_nonzero=True)
    if bad != 0:                                                            # assert!(sum.is_zero())
        raise MpcCheckError(f"SPDZ MAC check failed on {bad} of {n} opened values")
    return _after_consume(ctx, vals)


def gsz_batch_open(ctx, val: torch.Tensor, degree: int) -> torch.Tensor:
    """GszFieldShare::batch_open (mpc-algebra/src/share/gsz20/mod.rs:286-300): broadcast, per element the size-n_parties
    inverse DFT with the degree check, p(0)."""
    world, _ = _world_rank()
    n = val.shape[0]
    if _NET is not None and val.is_cuda:                                    # czk_gsz_batch_open
        val = val.contiguous()
        out = torch.empty_like(val)
        _net_in(ctx, val)
        bad = _NET.gsz_batch_open(val.data_ptr(), n, out.data_ptr(), degree=degree)
        if bad != 0:
            raise MpcCheckError(f"GSZ open: {bad} of {n} share polynomials exceed their degree bound")
        return _after_consume(ctx, out)
    gathered = all_gather_shares(val, ctx)
    out = torch.empty_like(val)
    bad = ctx.fr_gsz_open(gathered.data_ptr(), world, n, out.data_ptr(), degree=degree)
    if bad != 0:                                                            # assert!(p.degree() <= d)
        raise MpcCheckError(f"GSZ open: {bad} of {n} share polynomials exceed their degree bound")
    return _after_consume(ctx, out)


def gsz_batch_king_compute(ctx, val: torch.Tensor, degree: int) -> torch.Tensor:
    """gsz20::batch_king_compute(shares, new_degree, |r| r) (mpc-algebra/src/share/gsz20/mod.rs:494-527), the degree reduction inside batch_mult:
    every party's lane to the king, who opens each element with the degree bound and sends the VALUE back to every party as its new share."""
    world, rank = _world_rank()
    n = val.shape[0]
    if _NET is not None and val.is_cuda:                                    # czk_gsz_batch_king_compute
        val = val.contiguous()
        out = torch.empty_like(val)
        _net_in(ctx, val)
        bad = _NET.gsz_batch_king_compute(val.data_ptr(), n, out.data_ptr(), degree=degree)
        if bad != 0:
            raise MpcCheckError(f"GSZ king_compute: {bad} of {n} share polynomials exceed their degree bound")
        return _after_consume(ctx, out)
    _before_exchange(ctx, val)
    got = send_to_king(val)
    ans = None
    if rank == 0:
        opened = torch.empty_like(val)
        got = got.contiguous()
        bad = ctx.fr_gsz_open(got.data_ptr(), world, n, opened.data_ptr(), degree=degree)
        if bad != 0:
            raise MpcCheckError(f"GSZ king_compute: {bad} of {n} share polynomials exceed their degree bound")
        ctx.sync()
        ans = opened.unsqueeze(0).expand(world, *opened.shape)            # vec![output; n]
    return recv_from_king(ans, val)


def additive_batch_open(ctx, val: torch.Tensor) -> torch.Tensor:
    """AdditiveFieldShare::batch_open (mpc-algebra/src/share/add.rs:256-259; the reference's `--alg hbc`): broadcast the shares,
    sum them.  No MAC, no check.  val: (n, 4) device tensor."""
    world, _ = _world_rank()
    n = val.shape[0]
    if _NET is not None and val.is_cuda:                                    # czk_add_batch_open
        val = val.contiguous()
        out = torch.empty_like(val)
        _net_in(ctx, val)
        _NET.add_batch_open(val.data_ptr(), n, out.data_ptr())
        return _after_consume(ctx, out)
    gathered = all_gather_shares(val, ctx)
    out = torch.empty_like(val)
    ctx.fr_lanes_sum(gathered.contiguous().data_ptr(), world, n, out_ptr=out.data_ptr())
    return _after_consume(ctx, out)


def combine_split_results(ctx, czk, per_proof: list, keys=("h", "l", "a", "b_g1", "b_g2"), device=None):
    """Intra-party split (provers.Groth16Local base_split): every rank holds, per proof and query, the MSM over ITS base range as Jacobian
    limbs (lanes, 18 | 36).  One gather of all of them to rank 0, which adds the K partial sums of every (proof, query, lane) with the
    reference's GroupProjective::add_assign (czk_jac_add, host side: K - 1 additions per element -- "one extra point-add" per range).
    Returns the combined list on rank 0, None elsewhere."""
    import numpy as np
    world, rank = _world_rank()
    flat = np.concatenate([np.ascontiguousarray(r[k], dtype=np.uint64).reshape(-1) for r in per_proof for k in keys]) if per_proof else np.zeros(0, np.uint64)
    if world == 1:
        return per_proof
    t = torch.from_numpy(flat.view(np.int64).copy())
    if dist.get_backend() != "gloo":
        t = t.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    got = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
    dist.gather(t, got, dst=0)
    if rank != 0:
        return None
    parts = [g.cpu().numpy().view(np.uint64) for g in got]
    out, at = [], 0
    for r in per_proof:
        comb = {}
        for k in keys:
            lanes, jw = r[k].shape
            group = czk.CZK_G2 if jw == 36 else czk.CZK_G1
            acc = parts[0][at:at + lanes * jw].reshape(lanes, jw).copy()
            for p in parts[1:]:
                add = p[at:at + lanes * jw].reshape(lanes, jw)
                for ln in range(lanes):
                    acc[ln] = ctx.jac_add(group, acc[ln], add[ln])
            comb[k] = acc
            at += lanes * jw
        out.append(comb)
    return out
