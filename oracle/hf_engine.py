"""HF Transformers on the host CPU as a checker / CPU baseline (TEST INFRASTRUCTURE ONLY).

The reference has no CPU inference engine of its own: its CPU-capable path - and the engine that produced its golden
texts - is ``scripts/generate_test_data.py --device cpu``: ``AutoModelForCausalLM`` in bf16, ``model.generate(
do_sample=False, temperature=None, top_p=None)``, ``add_special_tokens=False`` (generate_test_data.py:60-95).  This module
runs exactly that engine and that ``generate`` call on an in-memory checkpoint (HF tensor names -> bf16 bits), so that

  * tests can pin the oracle AND the HIP path to the reference's truth engine on any seeded checkpoint - including a
    36-layer Qwen3-4B-shaped one, the configuration bench.py times - without real weights on disk;
  * bench.py's ``cpu_baseline`` leg can time "the reference's CPU path" (``kind: reference-engine``) on the host cores.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import time

import numpy as np


def _bits_to_bf16(bits, shape):
    import torch
    a = np.ascontiguousarray(bits, dtype=np.uint16).reshape(shape)
    return torch.from_numpy(a.view(np.int16)).view(torch.bfloat16)


def build_qwen3(cfg, tensor_bits, threads=None):
    """Qwen3ForCausalLM in bf16 holding exactly the given tensors (dict HF name -> uint16 bf16 bits; shapes are taken
    from the model).  No random initialisation is run (4 G parameters of normal_() would cost more than the forward)."""
    import torch
    from transformers import Qwen3Config, Qwen3ForCausalLM
    if threads:
        torch.set_num_threads(int(threads))
    keys = ["hidden_size", "num_hidden_layers", "num_attention_heads", "num_key_value_heads", "head_dim",
            "intermediate_size", "vocab_size", "rms_norm_eps", "tie_word_embeddings", "max_position_embeddings"]
    hf = Qwen3Config(**{k: cfg[k] for k in keys if k in cfg}, rope_theta=float(cfg.get("rope_theta", 1e6)),
                     attention_bias=False, use_sliding_window=False, attn_implementation="eager")
    try:   # skip the per-parameter normal_(): the tensors are assigned below
        try:
            from transformers.initialization import no_init_weights      # transformers >= 5
        except ImportError:
            from transformers.modeling_utils import no_init_weights      # 4.x
        ctx = no_init_weights()
    except Exception:  # noqa: BLE001 - older / newer layouts: pay for the init
        import contextlib
        ctx = contextlib.nullcontext()
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.bfloat16)
    try:
        with ctx:
            model = Qwen3ForCausalLM(hf)
    finally:
        torch.set_default_dtype(old)
    sd = model.state_dict()
    new = {}
    for name, ref in sd.items():
        src = name
        if name == "lm_head.weight" and name not in tensor_bits:
            src = "model.embed_tokens.weight"          # tied
        if src not in tensor_bits:
            raise KeyError(f"checkpoint has no tensor for {name}")
        new[name] = _bits_to_bf16(tensor_bits[src], tuple(ref.shape))
    model.load_state_dict(new, assign=True)
    if cfg.get("tie_word_embeddings", True):
        model.tie_weights()
    return model.eval()


class _Clock:
    """generate() streamer that stamps the arrival of every new token (the prompt arrives first, in one put())."""

    def __init__(self):
        self.t = []

    def put(self, value):
        self.t.append(time.perf_counter())

    def end(self):
        pass


def generate_greedy(model, prompt_tokens, max_new_tokens, return_logits=False):
    """The reference script's call (generate_test_data.py:78-86).  Returns (tokens, per-token wall-clock stamps with the
    stamp of the prompt hand-over first, logits [steps, vocab] float32 or None)."""
    import torch
    ids = torch.tensor([list(prompt_tokens)], dtype=torch.long)
    clock = _Clock()
    with torch.no_grad():
        out = model.generate(ids, max_new_tokens=int(max_new_tokens), do_sample=False, temperature=None, top_p=None,
                             top_k=None, pad_token_id=0, streamer=clock, output_logits=bool(return_logits),
                             return_dict_in_generate=True)
    toks = out.sequences[0, ids.shape[1]:].tolist()
    lg = torch.stack([x[0].float() for x in out.logits]).numpy() if return_logits else None
    return toks, clock.t, lg


def teacher_forced_logits(model, prompt_tokens, forced_tokens):
    """Logits of the last prompt position and of every decode step when the given tokens are fed (so HF, the oracle and
    the HIP path are compared on ONE token stream even where a near-tie would make their own greedy choices differ)."""
    import torch
    rows = []
    with torch.no_grad():
        out = model(input_ids=torch.tensor([list(prompt_tokens)], dtype=torch.long), use_cache=True)
        rows.append(out.logits[0, -1].float().numpy())
        past = out.past_key_values
        for t in forced_tokens:
            out = model(input_ids=torch.tensor([[int(t)]], dtype=torch.long), past_key_values=past, use_cache=True)
            past = out.past_key_values
            rows.append(out.logits[0, -1].float().numpy())
    return np.stack(rows)


def build_qwen35(cfg, tensors, threads=None):
    """Qwen3_5ForCausalLM (the hybrid linear / full attention text model) holding exactly the given tensors: dict of the
    reference's names (``model.language_model.*``, pegainfer-qwen35-4b/src/weights.rs:118) -> uint16 bf16 bits, or float32
    for ``A_log`` / ``linear_attn.norm.weight`` (weights.rs:226-241).  Same recipe as tests/golden/
    make_qwen35_tiny_golden.py (HF runs its torch fallbacks of the gated delta rule on CPU)."""
    import torch
    from transformers import Qwen3_5ForCausalLM, Qwen3_5TextConfig
    if threads:
        torch.set_num_threads(int(threads))
    keys = ["hidden_size", "intermediate_size", "num_hidden_layers", "vocab_size", "num_attention_heads",
            "num_key_value_heads", "head_dim", "linear_num_key_heads", "linear_num_value_heads", "linear_key_head_dim",
            "linear_value_head_dim", "linear_conv_kernel_dim", "rms_norm_eps", "layer_types"]
    hf = Qwen3_5TextConfig(**{k: cfg[k] for k in keys if k in cfg}, tie_word_embeddings=True,
                           max_position_embeddings=int(cfg.get("max_position_embeddings", 4096)), attention_bias=False,
                           rope_parameters=dict(rope_type="default", rope_theta=float(cfg.get("rope_theta", 1e7)),
                                                partial_rotary_factor=float(cfg.get("partial_rotary_factor", 0.25)),
                                                mrope_section=[11, 11, 10]),
                           attn_implementation="eager")
    try:
        try:
            from transformers.initialization import no_init_weights
        except ImportError:
            from transformers.modeling_utils import no_init_weights
        ctx = no_init_weights()
    except Exception:  # noqa: BLE001
        import contextlib
        ctx = contextlib.nullcontext()
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.bfloat16)
    try:
        with ctx:
            model = Qwen3_5ForCausalLM(hf)
    finally:
        torch.set_default_dtype(old)
    new = {}
    for name, ref in model.state_dict().items():
        src = "model.language_model." + (name[len("model."):] if name.startswith("model.") else name)
        if name == "lm_head.weight":
            src = "model.language_model.embed_tokens.weight"          # tied
        if src not in tensors:
            raise KeyError(f"checkpoint has no tensor for {name} ({src})")
        a = tensors[src]
        if a.dtype == np.float32:      # A_log / gated-norm weight: the reference keeps them f32 (weights.rs:226-241) and so
            # does this model - HF's forward upcasts A_log with .float() anyway; rounding a synthetic A_log to bf16 would
            # change every decay rate by up to 2^-9 and the recurrent state drifts over a 1024-token prompt
            new[name] = torch.from_numpy(np.ascontiguousarray(a)).reshape(tuple(ref.shape))
        else:
            new[name] = _bits_to_bf16(a, tuple(ref.shape))
    model.load_state_dict(new, assign=True)
    model.tie_weights()
    return model.eval()
