"""ctypes binding of the continuous-batching scheduler (include/pegainfer_scheduler.h).

``Scheduler.over_engine(engine)`` drives a Qwen3Engine through the C++ executor (the product path);
``Scheduler.over_callbacks(obj)`` runs the same C++ scheduling logic over a Python object implementing the executor
protocol of oracle/scheduler_ref.py - used by the CPU tests to replay the reference's FakeExecutor scenarios.
"""
import ctypes

import numpy as np

from . import ffi

TOKEN, FINISHED, ERROR, REJECTED, PROMPT_TOKEN = 1, 2, 3, 4, 5


class TokenEvent(ctypes.Structure):
    _fields_ = [("request_id", ctypes.c_uint64), ("kind", ctypes.c_int32), ("token", ctypes.c_uint32),
                ("finish_reason", ctypes.c_int32), ("prompt_tokens", ctypes.c_int32),
                ("completion_tokens", ctypes.c_int32), ("has_logprob", ctypes.c_int32), ("logprob", ctypes.c_float),
                ("n_top", ctypes.c_int32), ("top_index", ctypes.c_int32)]


_I32_V = ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p)
_STOP = ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p, ctypes.c_uint32)
_DROP = ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p, ctypes.c_uint64)
_EXEC = ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32,
                         ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_int32),
                         ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_float),
                         ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_float),
                         ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_uint32))
_ERR = ctypes.CFUNCTYPE(ctypes.c_void_p, ctypes.c_void_p)   # const char*: a pointer into a buffer we keep alive
_LP = ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32, ctypes.c_uint32, ctypes.c_int32,
                       ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_float))
_EXEC_ECHO = ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_uint64),
                              ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_uint32),
                              ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int32),
                              ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float),
                              ctypes.POINTER(ctypes.c_uint32))


class ExecutorVtbl(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_size_t), ("user", ctypes.c_void_p), ("page_size", _I32_V), ("max_request_pages", _I32_V),
                ("available_pages", _I32_V), ("is_stop_token", _STOP), ("drop_request", _DROP), ("execute", _EXEC),
                ("last_error", _ERR), ("max_batch_size", _I32_V), ("logprobs", _LP), ("execute_echo", _EXEC_ECHO),
                ("prompt_logprobs", _LP)]


class Scheduler:
    def __init__(self, handle, keep=None):
        self.lib = ffi.host_lib()
        self.h = handle
        self._keep = keep
        if not self.h:
            raise RuntimeError("scheduler creation failed")

    @classmethod
    def over_engine(cls, engine, seed=42, stop_tokens=()):
        lib = ffi.host_lib()
        st = np.ascontiguousarray(list(stop_tokens), dtype=np.uint32)
        create = lib.pegainfer_sched_create_qwen35 if type(engine).__name__ == "Qwen35Engine" \
            else lib.pegainfer_sched_create_qwen3
        h = create(engine.h, seed, st.ctypes.data if st.size else None, int(st.size))
        return cls(h, keep=engine)

    @classmethod
    def over_callbacks(cls, ex, seed=42, logprobs=True):
        lib = ffi.host_lib()
        state = {"buf": ctypes.create_string_buffer(512)}

        def execute(_u, n_pf, n_dec, ids, lens, tokens, temp, top_k, top_p, rv, out, echo=False):
            pf, dec, off = [], [], 0
            for i in range(n_pf + n_dec):
                params = (temp[i], top_k[i], top_p[i])
                if i < n_pf:
                    pf.append((ids[i], [tokens[off + j] for j in range(lens[i])], params, rv[i]))
                else:
                    dec.append((ids[i], tokens[off], params, rv[i]))
                off += lens[i]
            try:
                pt, dt = ex.execute(pf, dec, echo=True) if echo else ex.execute(pf, dec)
            except Exception as e:  # noqa: BLE001 - surfaced as the step's error message
                state["buf"].value = str(e).encode()[:511]
                return -1
            for i, t in enumerate(list(pt) + list(dt)):
                out[i] = int(t)
            return 0

        cbs = dict(page_size=_I32_V(lambda _u: ex.page_size()), max_request_pages=_I32_V(lambda _u: ex.max_request_pages()),
                   available_pages=_I32_V(lambda _u: ex.available_pages()),
                   is_stop_token=_STOP(lambda _u, t: 1 if ex.is_stop_token(t) else 0),
                   drop_request=_DROP(lambda _u, rid: (ex.drop_request(rid), 0)[1]), execute=_EXEC(execute),
                   last_error=_ERR(lambda _u: ctypes.addressof(state["buf"])))
        keys = ["page_size", "max_request_pages", "available_pages", "is_stop_token", "drop_request", "execute",
                "last_error"]
        size = ExecutorVtbl.max_batch_size.offset
        if hasattr(ex, "max_batch_size"):   # optional callback: NULL = unlimited
            cbs["max_batch_size"] = _I32_V(lambda _u: ex.max_batch_size())
            keys.append("max_batch_size")
            size = ExecutorVtbl.logprobs.offset
        if logprobs and hasattr(ex, "logprobs") and hasattr(ex, "prompt_logprobs"):   # optional trailing block (round 4)
            def lp_cb(fn):
                def cb(_u, a, tok, k, out_lp, out_ids, out_vals):
                    try:
                        lp, top = fn(int(a), int(tok), int(k))
                    except Exception as e:  # noqa: BLE001
                        state["buf"].value = str(e).encode()[:511]
                        return -1
                    out_lp[0] = lp
                    for i, (t, v) in enumerate(top[:k]):
                        out_ids[i], out_vals[i] = int(t), float(v)
                    return min(len(top), k)
                return _LP(cb)
            if "max_batch_size" not in cbs:
                cbs["max_batch_size"] = _I32_V()      # NULL
                keys.append("max_batch_size")
            cbs["logprobs"] = lp_cb(ex.logprobs)
            cbs["execute_echo"] = _EXEC_ECHO(lambda _u, n_pf, ids, lens, tokens, temp, top_k, top_p, rv, out:
                                             execute(_u, n_pf, 0, ids, lens, tokens, temp, top_k, top_p, rv, out, echo=True))
            cbs["prompt_logprobs"] = lp_cb(ex.prompt_logprobs)
            keys += ["logprobs", "execute_echo", "prompt_logprobs"]
            size = ctypes.sizeof(ExecutorVtbl)
        # struct_size = the bytes this caller fills: without the optional trailing callbacks the table ends earlier,
        # exactly what a caller built against an older header would pass
        vt = ExecutorVtbl(size, None, *[cbs[k] for k in keys])
        h = lib.pegainfer_sched_create(ctypes.addressof(vt), seed)
        return cls(h, keep=(vt, cbs, state, ex))

    def submit(self, prompt, max_tokens, params=(0.0, -1, 1.0, False), logprobs=0, echo=False):
        p = np.ascontiguousarray(prompt, dtype=np.uint32)
        return int(self.lib.pegainfer_sched_submit_ex(self.h, p.ctypes.data, int(p.size), int(max_tokens), float(params[0]),
                                                      int(params[1]), float(params[2]), int(bool(params[3])),
                                                      int(logprobs), int(bool(echo))))

    def cancel(self, rid):
        self.lib.pegainfer_sched_cancel(self.h, int(rid))

    def step(self):
        return int(self.lib.pegainfer_sched_step(self.h))

    def poll(self, max_events=4096):
        buf = (TokenEvent * max_events)()
        n = self.lib.pegainfer_sched_poll(self.h, ctypes.addressof(buf), max_events)
        n_top = self.lib.pegainfer_sched_poll_tops(self.h, None, None, 0)
        ids, vals = np.zeros(max(n_top, 1), np.uint32), np.zeros(max(n_top, 1), np.float32)
        if n_top:
            self.lib.pegainfer_sched_poll_tops(self.h, ids.ctypes.data, vals.ctypes.data, n_top)
        out = []
        for e in buf[:n]:   # same 8-tuple as oracle/scheduler_ref.py: (..., message, None | (logprob, [(id, logprob)]))
            lp = None
            if e.has_logprob:
                lp = (float(e.logprob), [(int(ids[e.top_index + i]), float(vals[e.top_index + i])) for i in range(e.n_top)])
            out.append((e.request_id, e.kind, e.token, e.finish_reason, e.prompt_tokens, e.completion_tokens, "", lp))
        return out

    def num_active(self):
        return int(self.lib.pegainfer_sched_num_active(self.h))

    def num_deferred(self):
        return int(self.lib.pegainfer_sched_num_deferred(self.h))

    def last_message(self):
        m = self.lib.pegainfer_sched_last_message(self.h)
        return m.decode() if m else ""

    def close(self):
        if self.h:
            self.lib.pegainfer_sched_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass
