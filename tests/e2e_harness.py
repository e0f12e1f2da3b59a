"""End-to-end greedy-text harness shared by tests/test_e2e_golden.py (real checkpoints, gated on
PEGAINFER_TEST_MODEL_PATH) and tests/test_gpu_model.py (the same code path exercised on the tiny committed
checkpoint with a word-level tokenizer, so the harness itself is known to run).  Mirrors
pegainfer-qwen3-4b/tests/e2e.rs:108-221 phase by phase."""
import json
import os


def load_tokenizer(model_path):
    from tokenizers import Tokenizer
    return Tokenizer.from_file(os.path.join(model_path, "tokenizer.json"))


def stop_tokens(model_path):
    """generation_config.json eos_token_id (scalar or list), fallback config.json (config.rs:97-111)."""
    for fn in ("generation_config.json", "config.json"):
        p = os.path.join(model_path, fn)
        if os.path.exists(p):
            e = json.load(open(p)).get("eos_token_id")
            if e is not None:
                return list(e) if isinstance(e, list) else [e]
    return []


def load_engine(model_path, **kw):
    cfg = json.load(open(os.path.join(model_path, "config.json")))
    cfg = cfg.get("text_config", cfg)
    kw.setdefault("num_kv_pages", 1024)
    kw.setdefault("max_batch_size", 8)
    if "layer_types" in cfg and "linear_attention" in cfg["layer_types"]:
        from pegainfer_amd.qwen35 import Qwen35Engine
        return Qwen35Engine(cfg, **kw).load_safetensors_native(model_path)
    from pegainfer_amd.qwen3 import Qwen3Engine
    cfg.setdefault("head_dim", cfg["hidden_size"] // cfg["num_attention_heads"])
    kw.setdefault("decode_mode", 1)
    return Qwen3Engine(cfg, **kw).load_safetensors_native(model_path)


def generate(sched, tok, prompt, max_tokens):
    """e2e.rs generate_tokens: submit, then drain TokenEvents until Finished."""
    from pegainfer_amd import scheduler as S
    ids = tok.encode(prompt, add_special_tokens=False).ids
    rid = sched.submit(ids, max_tokens)
    out, reason = [], None
    for _ in range(100000):
        sched.step()
        for (r, kind, token, fin, *_rest) in sched.poll():
            if r != rid:
                continue
            if kind == S.TOKEN:
                out.append(token)
            elif kind == S.FINISHED:
                reason = fin
            elif kind in (S.ERROR, S.REJECTED):
                raise AssertionError("generation failed: " + sched.last_message())
        if reason is not None:
            return out, reason
    raise AssertionError("scheduler did not finish the request")


def run_e2e(model_path, golden_path, **engine_kw):
    from pegainfer_amd import scheduler as S
    cases = json.load(open(golden_path))["cases"]
    tok = load_tokenizer(model_path)
    eng = load_engine(model_path, **engine_kw)
    sched = S.Scheduler.over_engine(eng, seed=42, stop_tokens=stop_tokens(model_path))
    # 1. greedy correctness
    for c in cases:
        ids, reason = generate(sched, tok, c["prompt"], c["max_new_tokens"])
        text = tok.decode(ids, skip_special_tokens=True)
        assert text, c["name"]
        if len(ids) >= c["max_new_tokens"]:
            assert reason == 1, (c["name"], reason)      # FinishReason::Length
        assert text == c["output"], (c["name"], text, c["output"])
    # 2. multi-request: every case again on the same engine, and identical (determinism)
    for c in cases:
        ids, _ = generate(sched, tok, c["prompt"], c["max_new_tokens"])
        assert tok.decode(ids, skip_special_tokens=True) == c["output"], c["name"]
    # 3. consumer drop, then the scheduler must still serve
    first = cases[0]["prompt"]
    rid = sched.submit(tok.encode(first, add_special_tokens=False).ids, 10)
    sched.cancel(rid)
    for _ in range(20):
        sched.step()
    ids, _ = generate(sched, tok, first, 5)
    assert tok.decode(ids, skip_special_tokens=True)
    sched.close()
    eng.close()
