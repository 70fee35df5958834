"""Schedule perturbation ("chaos") for the multi-stream pipelines.

Every ordering the library and the compiled hosts rely on must be an event wait or stream order -- never timing.  The lab build's option "chaos"
(csrc/core.hip chaos_point; CZK_CHAOS=<seed> in the environment of a lab-linked host) makes the timing hostile on purpose: at every stage boundary
-- the brackets around the digit sort, the accumulate kernels, the bucket reduction, the NTT passes and the polynomial kernels, the MSM result copy,
every put / get / generation flip of the shared-memory and hipIpc transports -- it first enqueues a spin kernel of random length (up to 400 us) on the
stage's stream and / or sleeps on the host (up to 300 us), and the MSM workspace ring hands out its slots in random order instead of round-robin.
Covered: the sort -> accumulate -> reduce chain and the NTT -> MSM hand-over of the Groth16 step (4 proofs pipelined), the mark / settle transcript
points, arena reuse and open -> combine of tools/polyvm_host.hpp (Plonk and Marlin, 4 proofs in flight on 4 contexts), and the mailbox generation flips of
csrc/net.hip (3 parties, one process each, hipIpc device mailboxes).  Every run must reproduce the unperturbed digest.  The last test REMOVES one event
wait (the reduce stream's wait for the accumulate kernel) and requires the same harness to notice."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(exe, argv, seed=None, drop=None, ranks=False, timeout=300):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    env.pop("CZK_CHAOS", None)
    env.pop("CZK_CHAOS_DROP_WAIT", None)
    if seed is not None:
        env["CZK_CHAOS"] = str(seed)
    if drop is not None:
        env["CZK_CHAOS_DROP_WAIT"] = str(drop)
    if ranks:
        return __import__("util").run_ranks([exe] + argv, capture_output=True, text=True, timeout=timeout, env=env)
    # (no core file, CPU or GPU: a deliberately broken schedule may end in a memory fault, and dumping a 288 GB device takes minutes)
    return subprocess.run([exe] + argv, capture_output=True, text=True, timeout=timeout, env=env,
                          preexec_fn=lambda: __import__("resource").setrlimit(__import__("resource").RLIMIT_CORE, (0, 0)))


def _line(r):
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


@pytest.mark.gpu
def test_perturbed_schedules_reproduce_the_digests():
    from test_abi import _build_host_demo
    lab = _build_host_demo(lab=True)
    cases = [("groth16, 4 proofs pipelined", ["bench", "--log-n", "13", "--steps", "4", "--warmup", "1"], "results_sha256", False),
             ("plonk, 4 in flight", ["plonk", "--log-n", "11", "--parties", "3", "--steps", "8", "--warmup", "1", "--inflight", "4"], "output_sha256", False),
             ("marlin, 4 in flight", ["marlin", "--log-n", "11", "--parties", "2", "--steps", "8", "--warmup", "1", "--inflight", "4"], "output_sha256", False),
             ("groth16, 3 parties over hipIpc mailboxes", ["party-launch", "--world", "3", "--transport", "ipc", "--log-n", "11", "--steps", "3", "--warmup", "1"],
              "results_sha256", True)]
    for what, argv, key, ranks in cases:
        want = _line(_run(lab, argv, ranks=ranks))[key]                    # the lab library, unperturbed
        for seed in (0x5EED1, 0x5EED2):
            j = _line(_run(lab, argv, seed=seed, ranks=ranks))
            assert j[key] == want, (what, hex(seed))
            if "in_flight_provers_equal" in j:
                assert j["in_flight_provers_equal"] is True


@pytest.mark.gpu
def test_the_product_library_gives_the_same_digests_and_ignores_the_switch():
    """libczk_hip.so reads no environment (tests/test_abi.py): CZK_CHAOS must change nothing -- and its digests are the lab library's."""
    from test_abi import _build_host_demo
    prod, lab = _build_host_demo(), _build_host_demo(lab=True)
    argv = ["plonk", "--log-n", "10", "--parties", "3", "--steps", "2", "--warmup", "1", "--inflight", "2"]
    a, b, c = _line(_run(prod, argv)), _line(_run(prod, argv, seed=7)), _line(_run(lab, argv, seed=7))
    assert a["output_sha256"] == b["output_sha256"] == c["output_sha256"]


@pytest.mark.gpu
def test_a_removed_event_wait_is_caught():
    """CZK_CHAOS_DROP_WAIT=1: the reduce stream no longer waits for its MSM's accumulate kernel (only for the digit sort), so the over-full-bucket items and
    the bucket reduction fold buckets that are stale or half written.  The breakage is confined to bucket CONTENTS -- every index structure the kernels read
    stays complete and protected, so the broken runs cannot fault or spin (the first version dropped the accumulate stream's wait for the sort instead: kernels
    then read another call's buffer layout as counts and indices, and one run in a few hung for minutes).  The harness must notice (a wrong digest, unequal
    pipelined proofs, a failed result check or a process that does not come back) for at least one of a few seeds -- and stops at the first that does."""
    from test_abi import _build_host_demo
    lab = _build_host_demo(lab=True)
    argv = ["plonk", "--log-n", "11", "--parties", "3", "--steps", "8", "--warmup", "1", "--inflight", "4"]
    want = _line(_run(lab, argv))["output_sha256"]
    caught = 0
    for seed in (11, 12, 13):
        try:
            r = _run(lab, argv, seed=seed, drop=1, timeout=60)
        except subprocess.TimeoutExpired:
            caught += 1
            break
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines or json.loads(lines[-1])["output_sha256"] != want:
            caught += 1
            break
    assert caught >= 1
