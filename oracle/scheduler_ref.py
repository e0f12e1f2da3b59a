"""CPU restatement of the reference's continuous-batching scheduler (TEST INFRASTRUCTURE ONLY).

Follows pegainfer-qwen3-4b/src/scheduler.rs:97-327 (loop, admission by KV-page budget, failure handling),
scheduler/plan.rs:31-117 (Prefill / Decode / Unified plan, one random_val per request per step, prompts first),
scheduler/resolve.rs:9-132 (stop token / length rules) and scheduler/effects.rs:67-217 (event order, swap_remove
retirement, dropped-receiver cleanup).  One call of ``step()`` = one iteration of ``scheduler_loop``; channels
become per-scheduler event lists.  logprobs / echo (executor.rs:211-284, resolve.rs:40-49,66,86,118,126,
effects.rs:75-82) are modelled since round 4: events carry an 8th element, None or (logprob, [(id, logprob), ...]).

Executor protocol (duck-typed, mirrors executor.rs:502-512): page_size(), max_request_pages(), available_pages(),
is_stop_token(tok), drop_request(id), execute(prefill_items, decode_items) -> (prefill_tokens, decode_tokens) or raises.
prefill_items: [(id, prompt, params, random_val)], decode_items: [(id, last_token, params, random_val)].
Optional, for logprobs / echo: logprobs(row, token, top_k) and prompt_logprobs(position, target, top_k) ->
(logprob, [(id, logprob), ...]) for the LAST execute (rows = requests, prompts first); execute(..., echo=True) is the
pure-Prefill plan with any_echo (plan.rs:62-66).
"""

TOKEN, FINISHED, ERROR, REJECTED, PROMPT_TOKEN = 1, 2, 3, 4, 5
STOP, LENGTH = 0, 1
PLAN_NONE, PLAN_PREFILL, PLAN_DECODE, PLAN_UNIFIED, STEP_FAILED = 0, 1, 2, 3, -1


def pages_needed(tokens, page_size):
    return -(-tokens // page_size)


from .std_rng import StdRng   # noqa: E402  the reference's StdRng::seed_from_u64 stream (restated; oracle/std_rng.py)


class SchedulerOracle:
    def __init__(self, executor, seed=42):
        self.ex = executor
        self.rng = StdRng(seed)
        self.active = []     # dicts: id, closed, last_token, generated, max_tokens, prompt_len, params
        self.deferred = []   # dicts: id, prompt, params, max_tokens, closed
        self.next_id = 0
        self.events = []     # (request_id, kind, token, finish_reason, prompt_tokens, completion_tokens, message)
        self.closed = set()

    # ---- EngineHandle::submit + receiver drop ----
    def submit(self, prompt, max_tokens, params=(0.0, -1, 1.0, False), logprobs=0, echo=False):
        rid = self.next_id
        self.next_id += 1
        self.deferred.append(dict(id=rid, prompt=list(prompt), params=tuple(params), max_tokens=int(max_tokens),
                                  logprobs=max(int(logprobs), 0), echo=bool(echo)))
        return rid

    def cancel(self, rid):
        self.closed.add(rid)

    def _send(self, rid, kind, token=0, reason=0, prompt_tokens=0, completion_tokens=0, message="", logprob=None):
        if rid in self.closed:
            return False
        self.events.append((rid, kind, token, reason, prompt_tokens, completion_tokens, message, logprob))
        return True

    def _row_logprob(self, row, token, top_k):
        """Some(extract_logprobs(..)) when the request asked for logprobs (executor.rs:222-226, 273-277)"""
        fn = getattr(self.ex, "logprobs", None)
        return fn(row, token, top_k) if (top_k > 0 and fn) else None

    # ---- scheduler.rs:175-261 ----
    def _admit(self):
        ps = self.ex.page_size()
        future = sum(max(0, pages_needed(a["prompt_len"] + max(a["max_tokens"] - 1, 0), ps)
                         - pages_needed(a["prompt_len"] + max(a["generated"] - 1, 0), ps)) for a in self.active)
        budget = max(0, self.ex.available_pages() - future)
        pending, still, rejected = [], [], []
        # optional executor.max_batch_size(): rows one execute() call may carry (active requests always decode)
        cap = getattr(self.ex, "max_batch_size", None)
        rows_left = max(0, cap() - len(self.active)) if cap else -1
        for req in self.deferred:
            need = pages_needed(len(req["prompt"]) + max(req["max_tokens"] - 1, 0), ps)
            if need > self.ex.max_request_pages():
                rejected.append(req)
            elif need <= budget and rows_left != 0:
                budget -= need
                rows_left -= 1 if rows_left > 0 else 0
                pending.append(req)
            else:
                still.append(req)
        self.deferred = still
        return pending, rejected

    def step(self):
        if not self.active and not self.deferred:
            return PLAN_NONE
        pending, rejected = self._admit()
        for req in rejected:
            mx = len(req["prompt"]) + max(req["max_tokens"] - 1, 0)
            self._send(req["id"], REJECTED, prompt_tokens=len(req["prompt"]),
                       message="request requires more KV pages than this model instance can provide: "
                               f"prompt_tokens={len(req['prompt'])}, max_context_tokens={mx}")
        have_active = bool(self.active)
        if pending and have_active:
            plan = PLAN_UNIFIED
        elif pending:
            plan = PLAN_PREFILL
        elif have_active:
            plan = PLAN_DECODE
        else:
            return PLAN_NONE
        # failure targets (scheduler.rs:263-284): decode targets first for Unified
        targets = []
        if plan in (PLAN_DECODE, PLAN_UNIFIED):
            targets += [(a["id"], a["prompt_len"], a["generated"]) for a in self.active]
        if plan in (PLAN_PREFILL, PLAN_UNIFIED):
            targets += [(p["id"], len(p["prompt"]), 0) for p in pending]
        pf_items = [(p["id"], p["prompt"], p["params"], self.rng.next_f32()) for p in pending] \
            if plan != PLAN_DECODE else []
        dec_items = [(a["id"], a["last_token"], a["params"], self.rng.next_f32()) for a in self.active] \
            if plan != PLAN_PREFILL else []
        any_echo = plan == PLAN_PREFILL and any(p["echo"] for p in pending)
        echo_step = any_echo and hasattr(self.ex, "prompt_logprobs")
        try:
            if echo_step:
                pf_tokens, dec_tokens = self.ex.execute(pf_items, dec_items, echo=True)
            else:
                pf_tokens, dec_tokens = self.ex.execute(pf_items, dec_items)
            # build_prefill_request_results / build_decode_request_results (executor.rs:211-284): part of the step
            pf_lps = [self._row_logprob(i, tok, p["logprobs"]) for i, (p, tok) in enumerate(zip(pending, pf_tokens))] \
                if plan != PLAN_DECODE else []
            echo_lps, off = [], 0
            for p in (pending if plan != PLAN_DECODE else []):
                row = None
                if p["echo"]:
                    row = [None] * len(p["prompt"])
                    if echo_step:
                        for j in range(1, len(p["prompt"])):
                            row[j] = self.ex.prompt_logprobs(off + j - 1, p["prompt"][j], p["logprobs"])
                echo_lps.append(row)
                off += len(p["prompt"])
            dec_lps = [self._row_logprob(len(pf_items) + j, tok, a["logprobs"])
                       for j, (a, tok) in enumerate(zip(self.active, dec_tokens))] if plan != PLAN_PREFILL else []
        except Exception as e:  # scheduler.rs:307-327
            for rid, pt, ct in targets:
                self._send(rid, ERROR, prompt_tokens=pt, completion_tokens=ct, message=str(e))
                self.ex.drop_request(rid)
            self.active = []
            return STEP_FAILED
        # ---- prompt echoes first (resolve.rs:40-49, effects.rs:75-82) ----
        for p, row in zip(pending if plan != PLAN_DECODE else [], echo_lps):
            if row is not None:
                for j, t in enumerate(p["prompt"]):
                    self._send(p["id"], PROMPT_TOKEN, token=t, prompt_tokens=j, completion_tokens=len(p["prompt"]),
                               logprob=row[j])
        # ---- resolve decode (resolve.rs:96-132) + apply (effects.rs:84-159) ----
        retire = []
        for ((rid, _, _, _), tok), lp in zip(zip(dec_items, dec_tokens), dec_lps):
            idx = next((i for i, a in enumerate(self.active) if a["id"] == rid), None)
            if idx is None:
                continue
            a = self.active[idx]
            completion = a["generated"] + 1
            ignore_eos = a["params"][3]
            if (not ignore_eos) and self.ex.is_stop_token(tok):
                self._send(rid, FINISHED, reason=STOP, prompt_tokens=a["prompt_len"], completion_tokens=completion)
                self.ex.drop_request(rid)
                retire.append(idx)
            elif completion >= a["max_tokens"]:
                if self._send(rid, TOKEN, token=tok, logprob=lp):
                    self._send(rid, FINISHED, reason=LENGTH, prompt_tokens=a["prompt_len"],
                               completion_tokens=completion)
                self.ex.drop_request(rid)
                retire.append(idx)
            else:
                if not self._send(rid, TOKEN, token=tok, logprob=lp):
                    self.ex.drop_request(rid)
                    retire.append(idx)
                else:
                    a["last_token"] = tok
                    a["generated"] = completion
        for i in reversed(retire):                         # Vec::swap_remove
            self.active[i] = self.active[-1]
            self.active.pop()
        # ---- resolve prefill (resolve.rs:31-94) + apply (effects.rs:164-216) ----
        for (p, tok), lp in zip(zip(pending if plan != PLAN_DECODE else [], pf_tokens), pf_lps):
            rid, plen, ignore_eos = p["id"], len(p["prompt"]), p["params"][3]
            if (not ignore_eos) and self.ex.is_stop_token(tok):
                self._send(rid, FINISHED, reason=STOP, prompt_tokens=plen, completion_tokens=0)
                self.ex.drop_request(rid)
            elif p["max_tokens"] <= 1:
                if self._send(rid, TOKEN, token=tok, logprob=lp):
                    self._send(rid, FINISHED, reason=LENGTH, prompt_tokens=plen, completion_tokens=1)
                self.ex.drop_request(rid)
            else:
                if self._send(rid, TOKEN, token=tok, logprob=lp):
                    self.active.append(dict(id=rid, last_token=tok, generated=1, max_tokens=p["max_tokens"],
                                            prompt_len=plen, params=p["params"], logprobs=p["logprobs"]))
                else:
                    self.ex.drop_request(rid)
        return plan

    def poll(self):
        ev, self.events = self.events, []
        return ev


class FakeExecutor:
    """scheduler.rs:343-505: page accounting only; prefill token = 100 + id, decode token = 200 + id."""

    def __init__(self, max_request_pages, page_size=16, fail_decode_once=False, stop_tokens=()):
        self.ps = page_size
        self.max_pages = max_request_pages
        self.avail = max_request_pages
        self.held = {}
        self.fail_decode_once = fail_decode_once
        self.dropped = []
        self.stop = set(stop_tokens)
        self.calls = []

    def page_size(self):
        return self.ps

    def max_request_pages(self):
        return self.max_pages

    def available_pages(self):
        return self.avail

    def is_stop_token(self, tok):
        return tok in self.stop

    def drop_request(self, rid):
        if rid in self.held:
            self.avail += pages_needed(self.held.pop(rid), self.ps)
        self.dropped.append(rid)

    def _ensure(self, rid, tokens):
        grow = pages_needed(tokens, self.ps) - pages_needed(self.held.get(rid, 0), self.ps)
        if grow > self.avail:
            raise RuntimeError("fake KV capacity exhausted")
        self.avail -= max(grow, 0)
        self.held[rid] = tokens

    # deterministic fake TokenLogprobs: a function of (row, token) resp. (position, target) only
    def logprobs(self, row, token, top_k):
        return (-0.5 - 0.001 * token - 0.25 * row, [(token + i, -1.0 - i - 0.125 * row) for i in range(top_k)])

    def prompt_logprobs(self, position, target, top_k):
        return (-0.25 * (position + 1), [(target + i, -2.0 - i) for i in range(top_k)])

    def execute(self, pf_items, dec_items, echo=False):
        self.calls.append((len(pf_items), len(dec_items)))
        self.echo_calls = getattr(self, "echo_calls", []) + [bool(echo)]
        if dec_items and not pf_items and self.fail_decode_once:
            self.fail_decode_once = False
            raise RuntimeError("fake decode KV capacity exhausted")
        for rid, prompt, _, _ in pf_items:
            self._ensure(rid, len(prompt))
        for rid, _, _, _ in dec_items:
            if rid not in self.held:
                raise RuntimeError("missing fake request state")
            self._ensure(rid, self.held[rid] + 1)
        return [100 + rid for rid, _, _, _ in pf_items], [200 + rid for rid, _, _, _ in dec_items]
