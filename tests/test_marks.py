"""czk_ctx_mark / czk_ctx_wait_mark / czk_lanes_download_deferred (include/czk.h): waiting for PART of a context's work.

A transcript point of the reference's polynomial provers waits for what the transcript has absorbed (mpc-plonk/src/lib.rs:430-448,
marlin/src/lib.rs:176-318), not for what the prover has started since.  Host results (czk_msm_async's points, deferred downloads) are
delivered by explicit calls only, so WHAT a wait delivers is deterministic: after wait_mark(m) the results enqueued before m are in the
caller's buffers and those enqueued after it are not -- whether or not the device has finished them."""
import numpy as np
import pytest

from util import rand_fr_canonical

SENTINEL = np.uint64(0xDEADBEEFDEADBEEF)


@pytest.mark.gpu
def test_wait_mark_delivers_what_was_enqueued_before_the_mark_and_nothing_else(orc):
    import czk_amd as czk
    ctx = czk.Context(0)
    n = 3000
    pts = ctx.fixed_base_points(czk.CZK_G1, rand_fr_canonical(11, n))
    bases = ctx.register_bases(czk.CZK_G1, pts, None)
    sc = [rand_fr_canonical(20 + i, n) for i in range(3)]
    want = [orc.jac_to_affine(1, orc.msm(1, pts, np.zeros(n, np.uint8), s))[0] for s in sc]
    dev = ctx.lanes_alloc(3, n)
    for i in range(3):
        dev.upload(sc[i], lane=i)
    outs = [np.full((1, 18), SENTINEL, dtype=np.uint64) for _ in range(3)]
    down = [np.full((5, 4), SENTINEL, dtype=np.uint64) for _ in range(2)]

    ctx.msm_async(bases, dev.ptr(0), n, 1, czk.CZK_SCALAR_CANONICAL, outs[0], stable=True)
    dev.download_deferred(down[0], lane=1, elem=7)
    m1 = ctx.mark()
    ctx.msm_async(bases, dev.ptr(1), n, 1, czk.CZK_SCALAR_CANONICAL, outs[1], stable=True)
    m2 = ctx.mark()
    ctx.msm_async(bases, dev.ptr(2), n, 1, czk.CZK_SCALAR_CANONICAL, outs[2], stable=True)
    dev.download_deferred(down[1], lane=2, elem=9)

    ctx.wait_mark(m1)
    assert np.array_equal(ctx.jac_to_affine(czk.CZK_G1, outs[0])[0][0], want[0])
    assert np.array_equal(down[0], sc[1][7:12])
    assert (outs[1] == SENTINEL).all() and (outs[2] == SENTINEL).all() and (down[1] == SENTINEL).all()   # delivered by explicit calls only
    ctx.wait_mark(m1)                                    # a retired mark: returns at once, delivers nothing
    assert (outs[1] == SENTINEL).all()
    ctx.wait_mark(m2)
    assert np.array_equal(ctx.jac_to_affine(czk.CZK_G1, outs[1])[0][0], want[1])
    assert (outs[2] == SENTINEL).all() and (down[1] == SENTINEL).all()
    ctx.sync()                                           # everything else
    assert np.array_equal(ctx.jac_to_affine(czk.CZK_G1, outs[2])[0][0], want[2])
    assert np.array_equal(down[1], sc[2][9:14])
    with pytest.raises(czk.CzkError):
        ctx.wait_mark(m2 + 100)                          # not a mark of this context
    dev.free()
    bases.release()
    ctx.close()


@pytest.mark.gpu
def test_a_later_mark_covers_the_earlier_ones_and_the_result_ring_wraps(orc):
    """wait_mark(m_k) without waiting for m_1 .. m_(k-1) delivers all of them; several thousand results go round the 4 MiB pinned ring
    (deferred downloads of 24 KiB each) without losing or reordering one."""
    import czk_amd as czk
    ctx = czk.Context(0)
    n = 768
    src = rand_fr_canonical(5, 64 * n)
    dev = ctx.lanes_alloc(1, 64 * n)
    dev.upload(src)
    rng = np.random.default_rng(6)
    got, marks, where = [], [], []
    for i in range(1200):                                 # 1200 x 24 KiB = 28 MiB through a 4 MiB ring
        at = int(rng.integers(0, 63 * n))
        o = np.full((n, 4), SENTINEL, dtype=np.uint64)
        dev.download_deferred(o, elem=at)
        got.append(o)
        where.append(at)
        if i % 7 == 6:
            marks.append((ctx.mark(), i))
        if i % 100 == 99:                                 # wait for the newest mark only
            m, upto = marks[-1]
            ctx.wait_mark(m)
            for k in range(upto + 1):
                assert np.array_equal(got[k], src[where[k]:where[k] + n]), k
            assert all((got[k] == SENTINEL).all() for k in range(upto + 1, i + 1))
    ctx.sync()
    for k in range(1200):
        assert np.array_equal(got[k], src[where[k]:where[k] + n]), k
    dev.free()
    ctx.close()
