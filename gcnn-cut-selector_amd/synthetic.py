"""Seeded synthetic (constraint, variable, cut) bipartite samples with the shapes of SURVEY.md section 8(d).

The reference ships no sample data (its `data/` directory is git-ignored), so benchmarks and parity tests run
on synthetic states laid out exactly as `utils.get_state` / `data_collector.py:135-140` produce them
(/root/reference/utils.py:35-238): per sample a 5-tuple of dicts
    (cons{'values'[C,4]}, cons_edge{'indices'[2,E1],'values'[E1,1]}, var{'values'[V,14]},
     cut{'values'[K,6]}, cut_edge{'indices'[2,E2],'values'[E2,1]})
plus a vector of K bound improvements.  Edges are emitted (row, col)-sorted like scipy's CSR->COO
(utils.py:102-104).  Sizes follow the reference's instance generators at LP-file level
(instance_generator.py:313-697); the cut side (K ~ U{10..100}, nnz ~ U{10..min(200,V)}) is an assumption.
Generator contract: numpy.random.default_rng(1000 * config_index + sample_index).
"""

from __future__ import annotations

import numpy as np

PROBLEMS = ("setcov", "combauc", "capfac", "indset")
CONFIG_INDEX = {"setcov": 1, "combauc": 2, "capfac": 3, "indset": 4}
FEATURE_NAMES = {
    "cons": ["rhs", "is_tight", "obj_cosine", "dual"],
    "edge": ["coef"],
    "var": ["type_0", "type_1", "type_2", "type_3", "obj_coef", "has_lb", "has_ub", "at_lb", "at_ub", "frac",
            "reduced_cost", "lp_val", "primal_val", "avg_primal"],
    "cut": ["rhs", "support", "int_support", "efficacy", "cutoff", "parallelism"],
}


def _rows_to_coo(row_cols):
    """list of sorted column arrays -> (row_idx, col_idx), (row, col)-sorted."""
    lens = np.fromiter((len(c) for c in row_cols), dtype=np.int64, count=len(row_cols))
    rows = np.repeat(np.arange(len(row_cols), dtype=np.int64), lens)
    cols = np.concatenate(row_cols) if len(row_cols) else np.zeros(0, np.int64)
    return rows, cols.astype(np.int64)


def _setcov_rows(rng, n_rows=500, n_cols=1000, nnz=25000, lo=25, hi=77):
    extra = rng.multinomial(nnz - lo * n_rows, np.full(n_rows, 1.0 / n_rows))
    counts = np.minimum(lo + extra, hi)
    deficit = nnz - int(counts.sum())
    while deficit > 0:  # redistribute what the cap removed
        room = np.flatnonzero(counts < hi)
        take = rng.choice(room, size=min(deficit, len(room)), replace=False)
        counts[take] += 1
        deficit = nnz - int(counts.sum())
    return [np.sort(rng.choice(n_cols, size=int(k), replace=False)) for k in counts], n_cols, -1.0


def _combauc_rows(rng, n_items=100, n_bids=500):
    # one row per item listing the bids that contain it (~192-200 rows once empty/duplicate rows are dropped,
    # ~2,700 nnz, row nnz 2/6/55) plus "dummy-item" rows tying a bidder's substitutable bids together
    weights = rng.pareto(1.5, n_items) + 0.05
    weights /= weights.sum()
    rows = [[] for _ in range(n_items)]
    for b in range(n_bids):
        size = int(min(n_items, 1 + rng.geometric(0.28)))
        for it in rng.choice(n_items, size=size, replace=False, p=weights):
            rows[it].append(b)
    n_dummy = int(rng.integers(92, 101))
    for _ in range(n_dummy):
        rows.append(list(rng.choice(n_bids, size=int(rng.integers(2, 6)), replace=False)))
    rows = [np.unique(np.asarray(r, np.int64)) for r in rows if len(r) >= 2]
    return rows, n_bids, 1.0


def _capfac_rows(rng, n_cust=100, n_fac=100):
    # vars: x_ij at i*n_fac + j (continuous), y_j at n_cust*n_fac + j (binary).  rows: 100 demand rows (100 nnz),
    # 100 capacity rows (101 nnz), 1 total-capacity row (100 nnz), 10,000 linking rows x_ij <= y_j (2 nnz).
    y0 = n_cust * n_fac
    rows = [np.arange(i * n_fac, (i + 1) * n_fac, dtype=np.int64) for i in range(n_cust)]
    rows += [np.concatenate([np.arange(j, y0, n_fac, dtype=np.int64), [y0 + j]]) for j in range(n_fac)]
    rows += [np.arange(y0, y0 + n_fac, dtype=np.int64)]
    rows += [np.array([i * n_fac + j, y0 + j], np.int64) for i in range(n_cust) for j in range(n_fac)]
    return rows, y0 + n_fac, 1.0


def _indset_rows(rng, n_nodes=750, affinity=4):
    # Barabasi-Albert preferential attachment; one <=1 row per edge, a few merged into 3-4 cliques.
    targets = list(range(affinity))
    repeated = []
    edges = []
    for new in range(affinity, n_nodes):
        for t in set(targets):
            edges.append((t, new))
        repeated.extend(targets)
        repeated.extend([new] * affinity)
        targets = [repeated[i] for i in rng.integers(0, len(repeated), size=affinity)]
        while len(set(targets)) < affinity:
            targets.append(repeated[int(rng.integers(0, len(repeated)))])
        targets = list(dict.fromkeys(targets))[:affinity]
    rows = [np.array(sorted(e), np.int64) for e in edges]
    for _ in range(len(rows) // 60):  # a handful of clique rows of size 3-4
        base = rows[int(rng.integers(0, len(rows)))]
        add = rng.choice(n_nodes, size=int(rng.integers(1, 3)), replace=False)
        rows.append(np.unique(np.concatenate([base, add])))
    return rows, n_nodes, 1.0


_BUILDERS = {"setcov": _setcov_rows, "combauc": _combauc_rows, "capfac": _capfac_rows, "indset": _indset_rows}


def make_sample(problem: str, sample_index: int, config_index: int | None = None, scale: float = 1.0):
    """One synthetic (state, improvements) pair in the reference's on-disk layout.

    `scale` < 1 shrinks the setcov/indset instance (tests only); 1.0 is the BASELINE size."""
    cfg = CONFIG_INDEX[problem] if config_index is None else config_index
    rng = np.random.default_rng(1000 * cfg + sample_index)
    if problem == "setcov" and scale != 1.0:
        nr, ncol = max(4, int(500 * scale)), max(8, int(1000 * scale))
        lo = max(2, int(25 * scale))
        rows, n_vars, sign = _setcov_rows(rng, nr, ncol, nnz=max(nr * lo, int(nr * ncol * 0.05)), lo=lo,
                                          hi=max(lo + 2, int(77 * scale) + 2))
    elif problem == "indset" and scale != 1.0:
        rows, n_vars, sign = _indset_rows(rng, max(12, int(750 * scale)))
    elif problem == "capfac" and scale != 1.0:
        n = max(3, int(100 * scale))
        rows, n_vars, sign = _capfac_rows(rng, n, n)
    else:
        rows, n_vars, sign = _BUILDERS[problem](rng)
    r_idx, c_idx = _rows_to_coo(rows)
    n_cons = len(rows)
    lens = np.array([len(r) for r in rows], dtype=np.float64)
    coef = (sign / np.sqrt(lens))[r_idx]  # +-1/sqrt(nnz_row): coefficient over row norm (utils.py:98-105)

    cons = np.stack([rng.standard_normal(n_cons), (rng.random(n_cons) < 0.3).astype(np.float64),
                     rng.uniform(-1, 1, n_cons), 0.1 * rng.standard_normal(n_cons)], axis=1)

    var = np.zeros((n_vars, 14))
    vtype = np.zeros(n_vars, dtype=np.int64)
    if problem == "capfac":
        vtype[: n_vars - int(round(np.sqrt(n_vars)))] = 3  # x_ij continuous, y_j binary
    var[np.arange(n_vars), vtype] = 1.0
    var[:, 4] = rng.standard_normal(n_vars)
    var[:, 5:9] = (rng.random((n_vars, 4)) < 0.5).astype(np.float64)
    var[:, 9] = rng.uniform(0, 0.5, n_vars)
    var[:, 10:14] = rng.standard_normal((n_vars, 4))

    n_cuts = int(rng.integers(10, 101))
    cut_rows, cut_vals = [], []
    for _ in range(n_cuts):
        nnz = int(rng.integers(10, min(201, n_vars + 1))) if n_vars >= 10 else int(rng.integers(1, n_vars + 1))
        cols = np.sort(rng.choice(n_vars, size=nnz, replace=False))
        vals = rng.standard_normal(nnz)
        cut_rows.append(cols)
        cut_vals.append(vals / np.linalg.norm(vals))
    k_idx, kc_idx = _rows_to_coo(cut_rows)
    cut = np.stack([rng.standard_normal(n_cuts), rng.random(n_cuts), rng.random(n_cuts),
                    np.abs(rng.standard_normal(n_cuts)), np.abs(rng.standard_normal(n_cuts)),
                    rng.uniform(-1, 1, n_cuts)], axis=1)
    improvements = rng.uniform(0, 0.1, n_cuts)

    state = ({"features": FEATURE_NAMES["cons"], "values": cons},
             {"features": FEATURE_NAMES["edge"], "indices": np.vstack([r_idx, c_idx]), "values": coef.reshape(-1, 1)},
             {"features": FEATURE_NAMES["var"], "values": var},
             {"features": FEATURE_NAMES["cut"], "values": cut},
             {"features": FEATURE_NAMES["edge"], "indices": np.vstack([k_idx, kc_idx]),
              "values": np.concatenate(cut_vals).reshape(-1, 1)})
    return state, improvements


def stack_samples(samples):
    """Disjoint-union batching of in-memory samples: the array half of `utils.load_batch`
    (/root/reference/utils.py:389-426).  Returns the 11-tuple with per-sample count vectors."""
    cons = [s[0][0]["values"] for s in samples]
    var = [s[0][2]["values"] for s in samples]
    cut = [s[0][3]["values"] for s in samples]
    n_cons = np.array([a.shape[0] for a in cons], np.int64)
    n_vars = np.array([a.shape[0] for a in var], np.int64)
    n_cuts = np.array([a.shape[0] for a in cut], np.int64)
    c_off = np.concatenate([[0], np.cumsum(n_cons)[:-1]])
    v_off = np.concatenate([[0], np.cumsum(n_vars)[:-1]])
    k_off = np.concatenate([[0], np.cumsum(n_cuts)[:-1]])
    cei = np.concatenate([s[0][1]["indices"] + np.array([[c_off[j]], [v_off[j]]]) for j, s in enumerate(samples)], 1)
    kei = np.concatenate([s[0][4]["indices"] + np.array([[k_off[j]], [v_off[j]]]) for j, s in enumerate(samples)], 1)
    return (np.concatenate(cons, 0).astype(np.float32), cei.astype(np.int32),
            np.concatenate([s[0][1]["values"] for s in samples], 0).astype(np.float32),
            np.concatenate(var, 0).astype(np.float32), np.concatenate(cut, 0).astype(np.float32),
            kei.astype(np.int32), np.concatenate([s[0][4]["values"] for s in samples], 0).astype(np.float32),
            n_cons.astype(np.int32), n_vars.astype(np.int32), n_cuts.astype(np.int32),
            np.concatenate([s[1] for s in samples]).astype(np.float32))


def make_batch(problem: str, batch_size: int, first_sample: int = 0, scale: float = 1.0):
    """Stacked synthetic mini-batch: (state10 with TOTAL counts as the model takes them, targets, per-sample n_cuts)."""
    b = stack_samples([make_sample(problem, first_sample + i, scale=scale) for i in range(batch_size)])
    state = b[:7] + (int(b[7].sum()), int(b[8].sum()), int(b[9].sum()))
    return state, b[10], b[9]
