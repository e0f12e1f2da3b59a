"""TEST INFRASTRUCTURE - CPU restatement of the reference's expert-parallel dispatch / combine semantics
(pegainfer-comm/src/ep_backend.rs:213-331: dispatch_send / dispatch_recv / combine_send / combine_recv; topology
pegainfer-comm/src/topology.rs: experts dealt to ranks in contiguous blocks).  Never imported by the product path.

The transport of the reference (pplx RDMA all-to-all) is an un-vendored dependency; what is restated here is the
contract its call sites rely on (pegainfer-comm/tests + the doc comments of ep_backend.rs):
  * dispatch: every (token, k) pair travels to the rank owning expert indices[t, k]; a rank receives its rows grouped
    by LOCAL expert; tokens_per_expert[e] = rows of local expert e.  Order inside an expert group: by source rank,
    then by (token, k) order on the source - the order the MI355X implementation guarantees (the reference leaves
    it unspecified beyond "grouped by expert").
  * combine: out[t] (+)= sum_k weights[t, k] * y(t, k), where y(t, k) is the expert output of that pair's row.
  * a pair whose index is not an expert id travels nowhere and contributes zero (defensive rule of this implementation;
    the reference does not define it).
Parity unpinned against reference golden vectors: the reference ships none for this path (its tests need 8 GPUs +
RDMA); pinned instead by the algebraic properties tested in tests/test_ep_oracle.py.
"""
import numpy as np


def dispatch(xs, indices, num_experts):
    """xs[r]: [T_r, H] float array of rank r; indices[r]: [T_r, topk] int.  Returns per rank
    (recv_rows [N_r, H], tokens_per_expert [E/W], origin [N_r, 3] = (src rank, token, k))."""
    world = len(xs)
    epr = num_experts // world
    out = []
    for dst in range(world):
        rows, origin, tpe = [], [], np.zeros(epr, np.uint32)
        for le in range(epr):
            e = dst * epr + le
            for src in range(world):
                idx = np.asarray(indices[src])
                idx = idx.reshape(len(xs[src]), idx.shape[-1] if idx.ndim > 1 else 1)
                for t in range(idx.shape[0]):
                    for k in range(idx.shape[1]):
                        if idx[t, k] == e:
                            rows.append(xs[src][t])
                            origin.append((src, t, k))
                            tpe[le] += 1
        h = xs[0].shape[1] if len(xs[0].shape) > 1 else 0
        out.append((np.asarray(rows, dtype=xs[0].dtype).reshape(len(rows), h), tpe, np.asarray(origin, np.int64).reshape(-1, 3)))
    return out


def combine(expert_rows, origins, weights, num_tokens, hidden, prev=None):
    """expert_rows[r]: [N_r, H] outputs in dispatch order of rank r; origins[r] from dispatch(); weights[r]: [T_r, topk].
    Returns out[r]: [T_r, H] float64 (prev[r] added when accumulating)."""
    world = len(expert_rows)
    outs = [np.zeros((num_tokens[r], hidden), np.float64) if prev is None else np.asarray(prev[r], np.float64).copy()
            for r in range(world)]
    ys = [dict() for _ in range(world)]
    for dst in range(world):
        for row, (src, t, k) in zip(expert_rows[dst], origins[dst]):
            ys[src][(int(t), int(k))] = np.asarray(row, np.float64)
    for r in range(world):
        w = np.asarray(weights[r], np.float64)
        w = w.reshape(num_tokens[r], w.shape[-1] if w.ndim > 1 else 1)
        for t in range(num_tokens[r]):
            for k in range(w.shape[1]):
                if (t, k) in ys[r]:      # a pair with an index outside [0, num_experts) was routed nowhere: zero
                    outs[r][t] += w[t, k] * ys[r][(t, k)]
    return outs
