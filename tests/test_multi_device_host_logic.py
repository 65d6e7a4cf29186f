"""Host logic of the one-process multi-device path (scri_amd/engine.py: `devices=`), without a GPU: how the time shards of a
pipelined call are dealt over the contexts, how one context's failure reaches the caller, how SCRI_AMD_DEVICES is read."""
import numpy as np
import pytest


class _FakeContext:
    def __init__(self, device):
        self.device = device
        self.handle = object()


def test_every_shard_is_dealt_exactly_once_in_contiguous_runs():
    from scri_amd import engine

    for n_ctx in (1, 2, 3, 4, 5, 8):
        for pieces in (1, 2, 3, 7, 10, 12, 24, 25):
            ctxs = [_FakeContext(0) for _ in range(n_ctx)]
            seen = []

            def call(ctx, p0, p1):
                seen.append((ctxs.index(ctx), p0, p1))

            errors = engine._run_dealt(ctxs, pieces, call)
            assert errors == [None] * n_ctx
            seen.sort(key=lambda x: x[1])
            covered = [p for _, p0, p1 in seen for p in range(p0, p1)]
            assert covered == list(range(pieces)), (n_ctx, pieces)  # once each, and a context's shards are one run
            assert [k for k, _, _ in seen] == sorted(k for k, _, _ in seen)  # earlier contexts take earlier times
            sizes = [p1 - p0 for _, p0, p1 in seen]
            assert max(sizes) - min(sizes + [pieces // n_ctx]) <= 1


def test_shard_counts_keep_three_shards_per_context():
    from scri_amd import engine

    assert engine.pieces_for([0]) == max(3, engine.PIPELINE_PIECES)
    assert engine.pieces_for([0, 0, 0, 0]) == 12 and engine.pieces_for(list(range(8))) == 24
    for n in range(1, 17):
        p = engine.pieces_for([0] * n)
        assert p % n == 0 and p // n >= 3 and p >= engine.PIPELINE_PIECES
    # with the series' shape: every context's share cut by the one-context rule, two shards per context at least
    row16 = 285 * 16
    assert engine.pieces_for(list(range(8)), 100_000, 16, 100_000 * row16) == 8 * 2  # cfg3: 12 500 rows per device, two shards of 6 250
    assert engine.pieces_for(list(range(8)), 1_000_000, 16, 1_000_000 * row16) == 8 * 20  # cfg4: 125 000 rows per device, the cap
    assert engine.pieces_for([0, 1], 100_000, 16, 100_000 * row16) == 2 * 8
    assert engine.pieces_for(list(range(8)), 200_000, None, 6 * 200_000 * 625 * 16, abd=True) == 8 * 10  # cfg5: 1.5 GB per device
    assert engine.pieces_for(list(range(8)), 16_000, None, 6 * 16_000 * 625 * 16, abd=True) == 8 * 2
    for n in (1, 2, 4, 8):
        for rows in (500, 20_000, 3_000_000):
            p = engine.pieces_for([0] * n, rows, 16, rows * row16)
            assert p % n == 0 and 1 <= p // n <= 20


def test_a_failing_context_reaches_the_caller():
    from scri_amd import engine

    ctxs = [_FakeContext(0) for _ in range(3)]

    def call(ctx, p0, p1):
        if ctx is ctxs[1]:
            raise ValueError("device 1: halo too small")

    errors = engine._run_dealt(ctxs, 9, call)
    assert errors[0] is None and isinstance(errors[1], ValueError) and errors[2] is None


def test_default_devices_from_the_environment(monkeypatch):
    from scri_amd import engine

    monkeypatch.delenv("SCRI_AMD_DEVICES", raising=False)
    assert engine.default_devices() is None
    monkeypatch.setenv("SCRI_AMD_DEVICES", "0, 2,3")
    assert engine.default_devices() == [0, 2, 3]
    monkeypatch.setenv("SCRI_AMD_DEVICES", "all")
    assert engine.default_devices() == [0] or len(engine.default_devices()) >= 1  # (no GPU here: one entry)


def test_dealt_rotation_blocks_cover_the_series_and_raise_the_first_error():
    from scri_amd import engine

    data = np.zeros((30001, 3), dtype=complex)
    seen = []
    fakes = [_FakeContext(0) for _ in range(5)]
    orig = engine.contexts_for
    engine.contexts_for = lambda devices, first=None: fakes
    try:
        engine._rotate_dealt(data, [0] * 5, None, lambda cx, r0, r1: seen.append((r0, r1)))
        seen.sort()
        assert seen[0][0] == 0 and seen[-1][1] == 30001 and all(a[1] == b[0] for a, b in zip(seen, seen[1:]))

        def failing(cx, r0, r1):
            if cx is fakes[3]:
                raise ValueError("row stride smaller than the modes")

        with pytest.raises(ValueError, match="row stride"):
            engine._rotate_dealt(data, [0] * 5, None, failing)
    finally:
        engine.contexts_for = orig


def test_devices_refuse_shards_and_device_pointers():
    from scri_amd import engine

    tr = engine.make_transformation(np.zeros(4, dtype=complex), [1, 0, 0, 0], [0, 0, 0], 9, 9, 2)
    t = np.arange(40.0)
    with pytest.raises(ValueError, match="devices"):
        engine.transform_modes(t, np.zeros((40, 5), dtype=complex), 2, 2, -2, -1, engine.BMS_TERM_H, tr, devices=[0, 0], shard=(0, 40, 0, 40))
    with pytest.raises(ValueError, match="devices"):
        engine.transform_abd(t, 0, 2, tr, devices=[0, 0], device=True, out_ptr=0)


def test_time_shard_count_of_the_host_path_follows_the_series_shape():
    """engine.auto_pieces: shards of at least 100 000 / l_max rows, at most 20, two from 16 MB on, one call below
    (profiles/r06_r_host_path_by_size.txt)"""
    from scri_amd import engine

    row16 = 285 * 16
    assert engine.auto_pieces(100_000, 16, 100_000 * row16) == 16  # cfg3
    assert engine.auto_pieces(1_000_000, 16, 1_000_000 * row16) == 20  # cfg4: the cap
    assert engine.auto_pieces(20_000, 16, 20_000 * row16) == 3
    assert engine.auto_pieces(10_000, 16, 10_000 * row16) == 2  # 43 MB: too short for two full shards, cut in two all the same
    assert engine.auto_pieces(2_000, 16, 2_000 * row16) == 1  # 9 MB: one call
    assert engine.auto_pieces(100_000, 8, 100_000 * 77 * 16) == 8
    assert engine.auto_pieces(2_000, 4, 2_000 * 21 * 16) == 1  # cfg1
    for n in (10, 1000, 10**5, 10**7):
        for L in (2, 8, 16, 32, 64):
            p = engine.auto_pieces(n, L, n * ((L + 1) ** 2 - 4) * 16)
            assert 1 <= p <= 20 and (p == 1 or n // p >= 1)
