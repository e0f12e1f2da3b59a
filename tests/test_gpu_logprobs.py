"""GPU: per-token logprobs / top_logprobs and prompt echo through the runtime and the scheduler C ABI.

executor.rs:400-434 (compute_logprobs_from_cpu), :807-831 (extract_logprobs / extract_prompt_logprobs), :211-284 (which
rows get one), resolve.rs:31-132 + effects.rs:75-82 (how they travel in the TokenEvent stream).  The tiny committed Qwen3
checkpoint is enough: the arithmetic is a host log-softmax over a logits row the engine already exposes - what is checked
is that the RIGHT row reaches it (request column, prompt position) and that the events carry it."""
import ctypes
import json
import os

import numpy as np
import pytest

from oracle import ops as O
from oracle.bf16 import bf16_from_bits

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def tiny(built_libs):
    from pegainfer_amd.qwen3 import Qwen3Engine
    meta = json.load(open(os.path.join(G, "qwen3_tiny_golden.json")))
    eng = Qwen3Engine(meta["config"], num_kv_pages=64, max_batch_size=4).load_safetensors(os.path.join(G, "qwen3_tiny.safetensors"))
    yield eng, meta
    eng.close()


def _engine_logprobs(eng, column, token, k):
    lp = ctypes.c_float(0)
    ids, vals = np.zeros(max(k, 1), np.uint32), np.zeros(max(k, 1), np.float32)
    n = eng.lib.pegainfer_qwen3_logprobs(eng.h, column, int(token), k, ctypes.addressof(lp), ids.ctypes.data, vals.ctypes.data)
    assert n >= 0, eng.lib.pegainfer_qwen3_last_error(eng.h)
    return lp.value, [(int(i), float(v)) for i, v in zip(ids[:n], vals[:n])]


def _same(a, b, tol=2e-5):
    return abs(a[0] - b[0]) <= tol and [t for t, _ in a[1]] == [t for t, _ in b[1]] and \
        all(abs(x - y) <= tol for (_, x), (_, y) in zip(a[1], b[1]))


def test_runtime_logprobs_rows_of_prefill_and_decode(tiny):
    eng, meta = tiny
    prompts = [meta["cases"][i]["prompt_tokens"] for i in (0, 2, 3)]
    rids = [eng.new_request() for _ in prompts]
    toks, lg = eng.prefill(rids, prompts, return_logits=True)
    for col in range(3):
        want = O.compute_logprobs(bf16_from_bits(lg[col]), int(toks[col]), 5)
        assert _same(_engine_logprobs(eng, col, toks[col], 5), want), col
    toks2, lg2 = eng.decode(rids, toks, return_logits=True)
    for col in range(3):
        other = int((int(toks2[col]) + 17) % eng.vocab)        # the logprob of a token that was NOT the argmax
        want = O.compute_logprobs(bf16_from_bits(lg2[col]), other, 3)
        assert _same(_engine_logprobs(eng, col, other, 3), want), col
    assert eng.lib.pegainfer_qwen3_logprobs(eng.h, 3, 0, 1, None, None, None) < 0       # no such column
    assert eng.lib.pegainfer_qwen3_logprobs(eng.h, 0, eng.vocab, 1, None, None, None) < 0   # token outside the vocabulary
    for r in rids:
        eng.drop_request(r)


def test_scheduler_events_carry_logprobs_and_prompt_echo(tiny):
    """The C++ scheduler over the real engine: an echo + logprobs request prefilled alone gets PromptTokens with
    [None, lp(1), ...] computed from the all-position logits (prefill.rs:196-212) and a TokenLogprob on every generated
    token - all equal to the oracle arithmetic on the logits the engine returns for the same prompts."""
    from pegainfer_amd.scheduler import PROMPT_TOKEN, TOKEN, Scheduler
    eng, meta = tiny
    prompt = meta["cases"][2]["prompt_tokens"]
    # expected values straight from the engine's own logits
    rid = eng.new_request()
    toks, last, allg = eng.prefill([rid], [prompt], echo=True)
    first = int(toks[0])
    want_echo = [None] + [O.compute_logprobs(bf16_from_bits(allg[j - 1]), prompt[j], 2) for j in range(1, len(prompt))]
    want_first = O.compute_logprobs(bf16_from_bits(last[0]), first, 2)
    t2, lg2 = eng.decode([rid], [first], return_logits=True)
    want_second = O.compute_logprobs(bf16_from_bits(lg2[0]), int(t2[0]), 2)
    eng.drop_request(rid)
    s = Scheduler.over_engine(eng)
    a = s.submit(prompt, 2, logprobs=2, echo=True)
    ev = []
    while s.step() != 0:
        ev += s.poll()
    ev += s.poll()
    echo = [e for e in ev if e[0] == a and e[1] == PROMPT_TOKEN]
    assert [e[2] for e in echo] == prompt and echo[0][7] is None
    for j in range(1, len(prompt)):
        assert _same(echo[j][7], want_echo[j]), j
    tok = [e for e in ev if e[0] == a and e[1] == TOKEN]
    assert [e[2] for e in tok] == [first, int(t2[0])]
    assert _same(tok[0][7], want_first) and _same(tok[1][7], want_second)
    # a request without logprobs gets none
    b = s.submit(prompt[:5], 2)
    ev = []
    while s.step() != 0:
        ev += s.poll()
    ev += s.poll()
    assert all(e[7] is None for e in ev if e[0] == b) and not [e for e in ev if e[1] == PROMPT_TOKEN]
    s.close()
