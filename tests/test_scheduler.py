"""Continuous-batching scheduler, CPU only.

The reference pins its scheduler with a FakeExecutor (pegainfer-qwen3-4b/src/scheduler.rs:330-733).  Those six
scenarios are restated here and run against BOTH the Python restatement (oracle/scheduler_ref.py) and the C++
scheduler in libpegainfer_qwen3.so driven through its callback executor, plus a randomized trace on which the two
must emit identical event streams and make identical executor calls.
"""
import numpy as np
import pytest

from oracle.scheduler_ref import (ERROR, FINISHED, LENGTH, REJECTED, STOP, TOKEN, FakeExecutor, SchedulerOracle,
                                  pages_needed)


def make(kind, ex, seed=42):
    if kind == "oracle":
        return SchedulerOracle(ex, seed)
    from pegainfer_amd.scheduler import Scheduler
    return Scheduler.over_callbacks(ex, seed)


def events_of(ev, rid):
    return [e[:6] for e in ev if e[0] == rid]


def run_until_idle(s, limit=1000):
    ev = []
    for _ in range(limit):
        plan = s.step()
        ev += [tuple(e[:6]) for e in s.poll()]
        if plan == 0:
            return ev
    raise AssertionError("scheduler did not go idle")


KINDS = ["oracle", "cxx"]


def test_kv_budget_counts_only_tokens_written_to_cache():  # scheduler.rs:507-546
    assert pages_needed(16 + max(1 - 1, 0), 16) == 1
    assert 16 + max(3 - 1, 0) == 18 and 16 + max(1 - 1, 0) == 16 and 16 + max(2 - 1, 0) == 17


@pytest.mark.parametrize("kind", KINDS)
def test_one_token_completion_on_page_boundary_fits_one_page(kind):  # scheduler.rs:548-569
    ex = FakeExecutor(1)
    s = make(kind, ex)
    rid = s.submit([1] * 16, 1)
    ev = run_until_idle(s)
    assert events_of(ev, rid) == [(rid, TOKEN, 100, 0, 0, 0), (rid, FINISHED, 0, LENGTH, 16, 1)]
    assert 0 in ex.dropped


@pytest.mark.parametrize("kind", KINDS)
def test_request_waits_for_full_kv_budget_before_prefill(kind):  # scheduler.rs:571-604
    ex = FakeExecutor(4)
    s = make(kind, ex)
    long_running = s.submit([1] * 16, 18)
    assert s.step() == 1 and events_of(s.poll(), long_running)[0][:3] == (long_running, TOKEN, 100)
    must_wait = s.submit([1] * 17, 1)
    order = []
    for _ in range(64):
        plan = s.step()
        for e in s.poll():
            order.append(tuple(e[:3]))
        if plan == 0:
            break
    first_wait = order.index((must_wait, TOKEN, 101))
    assert 0 in ex.dropped
    fin_long = next(i for i, e in enumerate(order) if e[0] == long_running and e[1] == FINISHED)
    assert fin_long < first_wait                      # admitted only after the first request released its KV
    assert (must_wait, FINISHED, 0) in order


@pytest.mark.parametrize("kind", KINDS)
def test_impossible_request_is_rejected_without_blocking_later_work(kind):  # scheduler.rs:640-671
    ex = FakeExecutor(2)
    s = make(kind, ex)
    too_large = s.submit([1] * 16, 34)
    s.step()
    ev = s.poll()
    assert [tuple(e[:6]) for e in ev] == [(too_large, REJECTED, 0, 0, 16, 0)]
    msg = ev[0][6] if kind == "oracle" else s.last_message()
    assert "requires more KV pages" in msg
    fits = s.submit([1] * 16, 1)
    ev = run_until_idle(s)
    assert events_of(ev, fits) == [(fits, TOKEN, 101, 0, 0, 0), (fits, FINISHED, 0, LENGTH, 16, 1)]


@pytest.mark.parametrize("kind", KINDS)
def test_decode_error_drops_request_state_and_scheduler_recovers(kind):  # scheduler.rs:673-716
    ex = FakeExecutor(4, fail_decode_once=True)
    s = make(kind, ex)
    will_fail = s.submit([1] * 16, 2)
    assert s.step() == 1
    assert [tuple(e[:3]) for e in s.poll()] == [(will_fail, TOKEN, 100)]
    assert s.step() == -1
    ev = s.poll()
    assert [tuple(e[:6]) for e in ev] == [(will_fail, ERROR, 0, 0, 16, 1)]
    msg = ev[0][6] if kind == "oracle" else s.last_message()
    assert "fake decode KV capacity exhausted" in msg
    assert 0 in ex.dropped
    after = s.submit([1] * 16, 1)
    ev = run_until_idle(s)
    assert events_of(ev, after) == [(after, TOKEN, 101, 0, 0, 0), (after, FINISHED, 0, LENGTH, 16, 1)]


@pytest.mark.parametrize("kind", KINDS)
def test_active_receiver_drop_releases_request_state(kind):  # scheduler.rs:718-733
    ex = FakeExecutor(4)
    s = make(kind, ex)
    rid = s.submit([1] * 16, 3)
    s.step()
    assert [tuple(e[:3]) for e in s.poll()] == [(rid, TOKEN, 100)]
    s.cancel(rid)
    run_until_idle(s)
    assert 0 in ex.dropped and ex.available_pages() == 4


@pytest.mark.parametrize("kind", KINDS)
def test_stop_token_finishes_without_emitting_it(kind):  # resolve.rs:49-58,110-116
    ex = FakeExecutor(8, stop_tokens=(200,))               # request 0's decode token is 200
    s = make(kind, ex)
    a = s.submit([1] * 5, 10)
    b = s.submit([1] * 5, 3, params=(0.0, -1, 1.0, True))  # ignore_eos: 201 is not a stop token anyway
    ev = run_until_idle(s)
    assert events_of(ev, a) == [(a, TOKEN, 100, 0, 0, 0), (a, FINISHED, 0, STOP, 5, 2)]
    assert events_of(ev, b) == [(b, TOKEN, 101, 0, 0, 0), (b, TOKEN, 201, 0, 0, 0), (b, TOKEN, 201, 0, 0, 0),
                                (b, FINISHED, 0, LENGTH, 5, 3)]


def test_randomized_trace_cxx_equals_oracle():
    """Random arrivals, prompt lengths, budgets, cancels and stop tokens: identical event streams, identical executor
    call sequence (plan kinds and batch compositions), identical drop order."""
    rng = np.random.default_rng(3)
    for trial in range(6):
        pages = int(rng.integers(6, 40))
        stops = (203, 207) if trial % 2 else ()
        exo, exc = FakeExecutor(pages, stop_tokens=stops), FakeExecutor(pages, stop_tokens=stops)
        so, sc = make("oracle", exo, 7), make("cxx", exc, 7)
        evo, evc = [], []
        for it in range(120):
            for _ in range(int(rng.integers(0, 3))):
                plen, mx = int(rng.integers(1, 90)), int(rng.integers(1, 40))
                params = (0.0, -1, 1.0, bool(rng.integers(0, 2)))
                assert so.submit([1] * plen, mx, params) == sc.submit([1] * plen, mx, params)
            if rng.random() < 0.05 and so.next_id:
                rid = int(rng.integers(0, so.next_id))
                so.cancel(rid)
                sc.cancel(rid)
            assert so.step() == sc.step()
            evo += [tuple(e[:6]) for e in so.poll()]
            evc += [tuple(e[:6]) for e in sc.poll()]
        assert evo == evc and len(evo) > 50
        assert exo.calls == exc.calls and exo.dropped == exc.dropped and exo.available_pages() == exc.available_pages()


class CappedExecutor(FakeExecutor):
    """FakeExecutor + the optional max_batch_size() callback; execute() fails like the model does when a call
    carries more rows than its decode buffers hold (qwen3_runtime.cpp decode()/prefill(): 'bad batch size')."""

    def __init__(self, pages, cap, **kw):
        super().__init__(pages, **kw)
        self.cap = cap

    def max_batch_size(self):
        return self.cap

    def execute(self, pf_items, dec_items):
        if len(pf_items) + len(dec_items) > self.cap:
            raise RuntimeError("bad batch size")
        return super().execute(pf_items, dec_items)


@pytest.mark.parametrize("kind", KINDS)
def test_admission_is_capped_by_the_executors_max_batch_size(kind):
    """admission went by KV pages only, so one request too many made execute() fail and
    fail_touched_requests dropped EVERY in-flight request.  With executor.max_batch_size() the scheduler never builds a
    call with more rows than the model takes: excess requests wait in `deferred` and are admitted as rows free up;
    nobody gets an ERROR."""
    ex = CappedExecutor(64, 3)
    s = make(kind, ex)
    rids = [s.submit([1] * 4, 3 + i) for i in range(7)]
    ev = run_until_idle(s)
    assert all(n_pf + n_dec <= 3 for n_pf, n_dec in ex.calls), ex.calls
    assert not [e for e in ev if e[1] == ERROR]
    for i, rid in enumerate(rids):
        mine = events_of(ev, rid)
        assert [e[1] for e in mine] == [TOKEN] * (3 + i) + [FINISHED] and mine[-1][3] == LENGTH
    assert ex.calls[0] == (3, 0)                                   # first iteration admits exactly the cap
    assert ex.available_pages() == 64 and sorted(ex.dropped) == rids


def test_capped_trace_cxx_equals_oracle():
    rng = np.random.default_rng(5)
    exo, exc = CappedExecutor(40, 4), CappedExecutor(40, 4)
    so, sc = make("oracle", exo, 9), make("cxx", exc, 9)
    evo, evc = [], []
    for it in range(150):
        for _ in range(int(rng.integers(0, 3))):
            plen, mx = int(rng.integers(1, 60)), int(rng.integers(1, 30))
            assert so.submit([1] * plen, mx) == sc.submit([1] * plen, mx)
        assert so.step() == sc.step()
        evo += [tuple(e[:6]) for e in so.poll()]
        evc += [tuple(e[:6]) for e in sc.poll()]
    assert evo == evc and exo.calls == exc.calls and max(a + b for a, b in exo.calls) <= 4
    assert not [e for e in evo if e[1] == ERROR]


def test_executor_vtbl_is_size_versioned():
    """pegainfer_executor_vtbl carries the caller's sizeof: a table that ends before the mandatory callbacks is
    rejected, and one that ends at last_error (a caller built before max_batch_size was appended) never has the
    bytes behind it read - here they hold a pointer that would crash if called."""
    import ctypes

    from pegainfer_amd import ffi
    from pegainfer_amd.scheduler import ExecutorVtbl, Scheduler
    lib = ffi.host_lib()
    ex = FakeExecutor(8)                       # no max_batch_size attribute -> the short table
    assert not hasattr(ex, "max_batch_size")
    s = Scheduler.over_callbacks(ex, logprobs=False)     # a caller built before the logprobs block existed either
    vt = s._keep[0]
    assert vt.struct_size == ExecutorVtbl.max_batch_size.offset < ExecutorVtbl.logprobs.offset < ctypes.sizeof(ExecutorVtbl)
    # poison every slot behind the declared size on a copy and create a second scheduler from it
    raw = (ctypes.c_char * ctypes.sizeof(ExecutorVtbl)).from_buffer_copy(vt)
    poisoned = ExecutorVtbl.from_buffer(raw)
    ctypes.memset(ctypes.addressof(poisoned) + ExecutorVtbl.max_batch_size.offset, 0x41,
                  ctypes.sizeof(ExecutorVtbl) - ExecutorVtbl.max_batch_size.offset)
    h = lib.pegainfer_sched_create(ctypes.addressof(poisoned), 42)
    assert h
    s2 = Scheduler(h, keep=(poisoned, raw, s))
    rid = s2.submit([1] * 16, 2, logprobs=3, echo=True)   # ... nor the logprobs / echo callbacks (no logprob comes back)
    ev = run_until_idle(s2)                     # admission consults max_batch_size when non-NULL: must not be called
    assert events_of(ev, rid)[-1][1] == FINISHED
    s2.close()
    # too short to hold the mandatory callbacks -> NULL
    poisoned.struct_size = ExecutorVtbl.execute.offset
    assert not lib.pegainfer_sched_create(ctypes.addressof(poisoned), 42)
    s.close()


# ---------------------------------------------------------------- logprobs / echo (round 4; executor.rs:211-284, 400-434)
def test_logprobs_from_logits_matches_the_oracle_restatement():
    """pegainfer_logprobs_from_logits (the C++ of compute_logprobs_from_cpu) == oracle.ops.compute_logprobs on rows with
    ties, top_k 0 / 1 / 5 / > n: ids identical (the reference's insertion order on ties), logprobs within 2 f32 ulp (expf)."""
    import ctypes
    from oracle import ops as O
    from pegainfer_amd import ffi
    lib = ffi.host_lib()
    rng = np.random.default_rng(11)
    for n, k in ((7, 0), (7, 1), (7, 5), (7, 9), (1024, 5), (151936, 20), (5, 3)):
        x = rng.standard_normal(n).astype(np.float32) * 3
        x = np.round(x * 4) / 4 if n <= 1024 else x           # coarse grid: exact ties
        tok = int(rng.integers(0, n))
        lp = ctypes.c_float(0)
        ids, vals = np.zeros(max(k, 1), np.uint32), np.zeros(max(k, 1), np.float32)
        got_n = lib.pegainfer_logprobs_from_logits(x.ctypes.data, n, tok, k, ctypes.addressof(lp), ids.ctypes.data,
                                                   vals.ctypes.data)
        want_lp, want_top = O.compute_logprobs(x, tok, k)
        assert got_n == len(want_top) == min(k, n)
        assert [int(i) for i in ids[:got_n]] == [t for t, _ in want_top], (n, k)
        assert abs(lp.value - want_lp) <= 4e-6 * max(1.0, abs(want_lp))
        assert np.allclose(vals[:got_n], [v for _, v in want_top], rtol=0, atol=4e-6 * max(1.0, abs(want_lp)))
    assert lib.pegainfer_logprobs_from_logits(x.ctypes.data, 0, 0, 1, None, None, None) == -1     # None in the reference
    assert lib.pegainfer_logprobs_from_logits(x.ctypes.data, 5, 5, 1, None, None, None) == -1


def _lp_close(a, b):
    if a is None or b is None:
        return a is None and b is None
    return abs(a[0] - b[0]) < 1e-6 and [t for t, _ in a[1]] == [t for t, _ in b[1]] and \
        all(abs(x - y) < 1e-6 for (_, x), (_, y) in zip(a[1], b[1]))


@pytest.mark.parametrize("kind", KINDS)
def test_token_logprobs_and_prompt_echo(kind):
    """One request with logprobs = 2 + echo prefilled alone (PromptTokens with logprobs [None, lp...], then every
    token with its TokenLogprob), one arriving while it decodes (Unified: echo comes back with None logprobs,
    executor.rs:352-358), one without logprobs (None everywhere)."""
    from oracle.scheduler_ref import PROMPT_TOKEN
    ex = FakeExecutor(20)
    s = make(kind, ex)
    a = s.submit([11, 12, 13], 3, logprobs=2, echo=True)
    assert s.step() == 1
    ev = s.poll()
    echo = [e for e in ev if e[1] == PROMPT_TOKEN]
    assert [(e[2], e[4], e[5]) for e in echo] == [(11, 0, 3), (12, 1, 3), (13, 2, 3)]
    assert echo[0][7] is None and _lp_close(echo[1][7], (-0.25, [(12, -2.0), (13, -3.0)]))
    assert _lp_close(echo[2][7], (-0.5, [(13, -2.0), (14, -3.0)]))
    tok = [e for e in ev if e[1] == TOKEN][0]
    assert tok[2] == 100 + a and _lp_close(tok[7], ex.logprobs(0, 100 + a, 2))
    assert ex.echo_calls == [True]
    b = s.submit([21, 22], 2, logprobs=1, echo=True)      # admitted next to the active request: Unified
    c = s.submit([31], 2)
    assert s.step() == 3
    ev = s.poll()
    echo_b = [e for e in ev if e[1] == PROMPT_TOKEN and e[0] == b]
    assert [e[2] for e in echo_b] == [21, 22] and all(e[7] is None for e in echo_b)
    assert ex.echo_calls == [True, False]
    order = [(e[0], e[1]) for e in ev]
    assert order.index((b, PROMPT_TOKEN)) < order.index((a, TOKEN)) < order.index((b, TOKEN))   # echoes first, decode, prefill
    tb = [e for e in ev if e[0] == b and e[1] == TOKEN][0]
    assert _lp_close(tb[7], ex.logprobs(0, 100 + b, 1))
    tc = [e for e in ev if e[0] == c and e[1] == TOKEN][0]
    assert tc[7] is None
    ta = [e for e in ev if e[0] == a and e[1] == TOKEN][0]
    assert _lp_close(ta[7], ex.logprobs(2, 200 + a, 2))   # decode rows follow the two prompts
    run_until_idle(s)


def test_logprobs_trace_cxx_equals_oracle():
    """Randomized arrivals with random logprobs / echo settings: the C++ scheduler and the oracle emit the same events
    with the same TokenLogprobs and make the same executor calls (incl. which steps asked for echo logits)."""
    rng = np.random.default_rng(5)
    for trial in range(4):
        pages = int(rng.integers(8, 40))
        exo, exc = FakeExecutor(pages), FakeExecutor(pages)
        so, sc = make("oracle", exo, 9), make("cxx", exc, 9)
        evo, evc = [], []
        for it in range(80):
            for _ in range(int(rng.integers(0, 3))):
                plen, mx = int(rng.integers(1, 60)), int(rng.integers(1, 20))
                lpk, echo = int(rng.integers(0, 4)), bool(rng.integers(0, 2))
                assert so.submit([7] * plen, mx, logprobs=lpk, echo=echo) == sc.submit([7] * plen, mx, logprobs=lpk, echo=echo)
            assert so.step() == sc.step()
            evo += so.poll()
            evc += sc.poll()
        assert [e[:6] for e in evo] == [e[:6] for e in evc] and len(evo) > 50
        assert all(_lp_close(a[7], b[7]) for a, b in zip(evo, evc))
        assert any(e[7] is not None for e in evo) and any(e[1] == 5 for e in evo)
        assert exo.calls == exc.calls and exo.echo_calls == exc.echo_calls
