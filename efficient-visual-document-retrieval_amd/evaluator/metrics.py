"""Retrieval metrics with trec_eval semantics (host side, numpy).

The reference delegates this to third-party `mteb` -> `pytrec_eval` (evaluator/retrieval.py:218,239-246),
neither installed nor version-pinned anywhere in the reference: PARITY UNPINNED at this boundary
(SURVEY §8(c)).  Semantics restated from trec_eval / mteb 1.x behaviour:
  * only queries present in both qrels and results are scored;
  * ranking: score descending, ties by docid string descending (trec_eval's comparator);
  * ndcg_cut: linear gain rel / log2(rank+1), ideal from the judged docs sorted by relevance;
  * recall_k = hits@k / #relevant, P_k = hits@k / k, map_cut_k = sum of precisions at relevant ranks <= k / #relevant;
  * MRR@k follows mteb's evaluate_custom("mrr"): plain score sort, first relevant rank, mean over len(qrels);
  * means are rounded to 5 decimals as mteb does.

Two entries, one arithmetic: `evaluate(qrels, results, ks)` takes the reference's dict-of-dicts (the compatibility entry
behind `compute_mteb_metrics`), `evaluate_topk(index, top_scores, top_idx, extra)` takes the device's top-k arrays and
never builds a per-query dict -- sums run in the same (sequential) order in both, so the two agree to the last bit, not
only to the rounded digit (tests/test_host_logic.py, tests/test_gpu_edge_semantics.py).
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np


def _order_trec(docids: np.ndarray, scores: np.ndarray) -> np.ndarray:
    # primary: score desc; secondary: docid desc.  lexsort sorts ascending by the LAST key first.
    rank_of_id = np.empty(len(docids), dtype=np.int64)
    rank_of_id[np.argsort(docids, kind="stable")] = np.arange(len(docids))
    return np.lexsort((-rank_of_id, -scores))


def _seq_sum(x: np.ndarray) -> float:
    """Left-to-right sum (np.sum is pairwise): the order the array path's row-wise cumsum uses."""
    return float(np.cumsum(x)[-1]) if len(x) else 0.0


def evaluate(qrels: Dict[str, Dict[str, int]], results: Dict[str, Dict[str, float]], k_values: Sequence[int]):
    ks = list(k_values)
    kmax = max(ks)
    disc = 1.0 / np.log2(np.arange(2, kmax + 2))
    acc = {name: np.zeros(len(ks)) for name in ("ndcg", "map", "recall", "p", "mrr")}
    n_eval = 0
    for qid, doc_scores in results.items():
        judged = qrels.get(qid)
        if judged is None:
            continue
        n_eval += 1
        ids = np.array(list(doc_scores.keys()), dtype=object).astype(str)
        sc = np.fromiter(doc_scores.values(), dtype=np.float64, count=len(doc_scores))
        order = _order_trec(ids, sc)[:kmax]
        gains = np.array([float(judged.get(d, 0)) for d in ids[order]])
        gains = np.where(gains > 0, gains, 0.0)
        pos_rels = np.sort(np.array([r for r in judged.values() if r > 0], dtype=np.float64))[::-1]
        nrel = len(pos_rels)
        hits = (gains > 0).astype(np.float64)
        plain = np.argsort(-sc, kind="stable")[:kmax]            # mteb mrr: stable sort on score only
        plain_hits = np.array([judged.get(d, 0) > 0 for d in ids[plain]])
        first = int(np.argmax(plain_hits)) if plain_hits.any() else -1
        for j, k in enumerate(ks):
            g = gains[:k]
            dcg = _seq_sum(g * disc[: len(g)])
            ideal = pos_rels[:k]
            idcg = _seq_sum(ideal * disc[: len(ideal)])
            acc["ndcg"][j] += dcg / idcg if idcg > 0 else 0.0
            h = hits[:k]
            nh = float(h.sum())
            acc["recall"][j] += nh / nrel if nrel else 0.0
            acc["p"][j] += nh / k
            if nrel and nh > 0:
                prec_at = np.cumsum(h) / np.arange(1, len(h) + 1)
                acc["map"][j] += _seq_sum(prec_at * h) / nrel
            if 0 <= first < k:
                acc["mrr"][j] += 1.0 / (first + 1)
    den = max(n_eval, 1)
    den_mrr = max(len(qrels), 1)
    out = {
        "NDCG": {f"NDCG@{k}": round(acc["ndcg"][j] / den, 5) for j, k in enumerate(ks)},
        "mAP": {f"MAP@{k}": round(acc["map"][j] / den, 5) for j, k in enumerate(ks)},
        "Recall": {f"Recall@{k}": round(acc["recall"][j] / den, 5) for j, k in enumerate(ks)},
        "Precision": {f"P@{k}": round(acc["p"][j] / den, 5) for j, k in enumerate(ks)},
        "mRR": {f"MRR@{k}": round(acc["mrr"][j] / den_mrr, 5) for j, k in enumerate(ks)},
    }
    return out


def results_from_topk(top_scores: np.ndarray, top_idx: np.ndarray, query_keys: Sequence[str],
                      docids: Sequence[str], extra=None) -> Dict[str, Dict[str, float]]:
    """Device top-k lists -> the {qid: {docid: score}} dict the metric wrapper consumes.  Replaces the
    reference's per-element `.item()` loop (mainv2_iter_distill_infonce.py:311-317) with two D2H copies.
    `extra` (ops.topk_with_ties): rows whose k-th score is tied beyond the device cut carry every column with a score >=
    the k-th one, so that the docid-descending tie rule of trec_eval picks from the same candidates as the reference's
    all-pairs dict does."""
    out: Dict[str, Dict[str, float]] = {}
    for qi, qk in enumerate(query_keys):
        row = {}
        if extra and qi in extra:
            cols, sc = extra[qi]
            for s, i in zip(sc.tolist(), cols.tolist()):
                row[docids[i]] = float(s)
        else:
            for s, i in zip(top_scores[qi].tolist(), top_idx[qi].tolist()):
                if i >= 0:
                    row[docids[i]] = float(s)
        out[str(qk)] = row
    return out


# ---- the array path: device top-k -> metric tables, no per-query dicts -------------------------------------------------
class EvalIndex:
    """What one (qrels, query keys, docids) triple contributes to every evaluation of a run, built ONCE: the docid ->
    descending-string-order rank that trec_eval's tie rule needs (as an integer key), the judged (query row, page column,
    relevance) triples, each query's ideal gain vector and number of relevant pages.  `usable` is False when query keys or
    docids repeat (a dict would merge them): callers then go through the dict entry."""

    def __init__(self, qrels: Dict[str, Dict[str, int]], query_keys: Sequence[str], docids: Sequence[str],
                 k_values: Sequence[int]):
        self.qrels = qrels
        self.ks = [int(k) for k in k_values]
        self.kmax = max(self.ks)
        self.query_keys = [str(q) for q in query_keys]
        self.docids = list(docids)
        nq, n = len(self.query_keys), len(self.docids)
        ids = np.asarray([str(d) for d in self.docids], dtype=object).astype(str) if n else np.zeros(0, dtype=str)
        self.docrank = np.empty(n, dtype=np.int64)
        self.docrank[np.argsort(ids, kind="stable")] = np.arange(n)
        col_of = {d: j for j, d in enumerate(self.docids)}
        self.usable = len(col_of) == n and len(set(self.query_keys)) == nq
        self.n_qrels = len(qrels)
        self.judged = np.zeros(nq, dtype=bool)
        self.nrel = np.zeros(nq, dtype=np.float64)
        self.ideal = np.zeros((nq, self.kmax), dtype=np.float64)
        jrow, jcol, jrel = [], [], []
        for i, qk in enumerate(self.query_keys):
            judged = qrels.get(qk)
            if judged is None:
                continue
            self.judged[i] = True
            pos = np.sort(np.array([r for r in judged.values() if r > 0], dtype=np.float64))[::-1]
            self.nrel[i] = len(pos)
            self.ideal[i, : min(len(pos), self.kmax)] = pos[: self.kmax]
            for d, r in judged.items():
                j = col_of.get(d)
                if j is not None and r > 0:
                    jrow.append(i)
                    jcol.append(j)
                    jrel.append(float(r))
        self.jrow = np.asarray(jrow, dtype=np.int64)
        self.jcol = np.asarray(jcol, dtype=np.int64)
        self.jrel = np.asarray(jrel, dtype=np.float64)
        disc = 1.0 / np.log2(np.arange(2, self.kmax + 2))
        self.disc = disc
        self.idcg = np.cumsum(self.ideal * disc[None, :], axis=1)             # (nq, kmax): ideal DCG at every cut-off


def _gains(index: EvalIndex, rows: np.ndarray, cols: np.ndarray) -> np.ndarray:
    """(m, w) gain matrix of candidate columns `cols` for query rows `rows`: one comparison of the judged triples of
    those rows against their candidate lists (J x w booleans; J = judged pairs, 1 per query on ViDoRe)."""
    g = np.zeros(cols.shape, dtype=np.float64)
    if len(index.jrow) == 0 or cols.size == 0:
        return g
    local = np.full(len(index.query_keys), -1, dtype=np.int64)
    local[rows] = np.arange(len(rows))
    lr = local[index.jrow]
    keep = lr >= 0
    lr, jc, jr = lr[keep], index.jcol[keep], index.jrel[keep]
    a, pos = np.nonzero(cols[lr] == jc[:, None])
    g[lr[a], pos] = jr[a]
    return g


def _per_query(index: EvalIndex, rows: np.ndarray, sc: np.ndarray, cols: np.ndarray, presorted: bool) -> np.ndarray:
    """(5, m, len(ks)) per-query ndcg / map / recall / p / mrr of candidate lists (m, w): scores `sc` fp32, page columns
    `cols` (< 0 = no candidate).  presorted: the lists are the device's top-k (score desc, index asc) -- only rows that
    hold equal neighbours, a NaN or a hole are re-ranked by trec_eval's rule; otherwise every row is."""
    m, w = sc.shape
    kmax, ks = index.kmax, index.ks
    if w < kmax:                                                  # fewer candidates than the largest cut-off: pad with holes
        sc = np.concatenate([sc, np.full((m, kmax - w), -np.inf, dtype=sc.dtype)], axis=1)
        cols = np.concatenate([cols, np.full((m, kmax - w), -1, dtype=cols.dtype)], axis=1)
    hole = cols < 0
    if presorted:
        redo = (sc[:, 1:] == sc[:, :-1]).any(axis=1) | np.isnan(sc).any(axis=1) | hole.any(axis=1)
    else:
        redo = np.ones(m, dtype=bool)
    trec_cols = cols[:, :kmax].copy()
    plain_cols = cols[:, :kmax].copy()
    if redo.any():
        r = np.nonzero(redo)[0]
        s_r, c_r, h_r = sc[r].astype(np.float64), cols[r], hole[r]
        rank = np.where(h_r, 0, index.docrank[np.where(h_r, 0, c_r)])
        order = np.lexsort((-rank, -s_r, h_r), axis=-1)[:, :kmax]     # holes last, score desc, docid desc
        trec_cols[r] = np.take_along_axis(c_r, order, axis=1)
        key = np.where(h_r, np.nan, -s_r)                             # mteb's mrr: stable sort on the score alone (NaN last)
        plain = np.argsort(key, axis=1, kind="stable")[:, :kmax]
        plain_cols[r] = np.take_along_axis(c_r, plain, axis=1)
    g = _gains(index, rows, trec_cols)
    hits = (g > 0).astype(np.float64)
    disc = index.disc
    dcg = np.cumsum(g * disc[None, :], axis=1)
    cum = np.cumsum(hits, axis=1)
    ap = np.cumsum(cum / np.arange(1, kmax + 1)[None, :] * hits, axis=1)
    ph = _gains(index, rows, plain_cols) > 0
    first = np.where(ph.any(axis=1), ph.argmax(axis=1), kmax)
    nrel = index.nrel[rows]
    safe = np.where(nrel > 0, nrel, 1.0)
    out = np.zeros((5, m, len(ks)), dtype=np.float64)
    for j, k in enumerate(ks):
        idcg = index.idcg[rows, k - 1]
        out[0, :, j] = np.where(idcg > 0, dcg[:, k - 1] / np.where(idcg > 0, idcg, 1.0), 0.0)
        out[1, :, j] = np.where(nrel > 0, ap[:, k - 1] / safe, 0.0)
        out[2, :, j] = np.where(nrel > 0, cum[:, k - 1] / safe, 0.0)
        out[3, :, j] = cum[:, k - 1] / k
        out[4, :, j] = np.where(first < k, 1.0 / (first + 1.0), 0.0)
    return out


def evaluate_topk(index: EvalIndex, top_scores: np.ndarray, top_idx: np.ndarray, extra=None):
    """The five metric tables straight from the device's (nq, k) top-k arrays (+ `extra` of `ops.topk_with_ties`): the same
    numbers as `evaluate(qrels, results_from_topk(...), ks)`, bit for bit, without a dict per query -- gains come from one
    array comparison, the trec_eval tie rule is a lexsort on (-score, -docid rank) applied only to rows that have ties, and
    every metric at every cut-off is a cumulative sum along the rank axis.  Needs k >= max(k_values) for the reference's
    cut-offs to see k candidates (the drivers use k = 100 = max of evaluator/retrieval.py:223)."""
    if not index.usable:                                          # repeated keys: a dict merges them, so let the dict entry decide
        return evaluate(index.qrels, results_from_topk(np.asarray(top_scores), np.asarray(top_idx), index.query_keys,
                                                       index.docids, extra=extra), index.ks)
    ks = index.ks
    nq = len(index.query_keys)
    ts = np.asarray(top_scores)
    ti = np.asarray(top_idx).astype(np.int64)
    vals = np.zeros((5, nq, len(ks)), dtype=np.float64)
    rows = np.nonzero(index.judged)[0]
    tie_rows = np.array(sorted(r for r in (extra or {}) if index.judged[r]), dtype=np.int64)
    plain_rows = np.setdiff1d(rows, tie_rows) if len(tie_rows) else rows
    if len(plain_rows):
        vals[:, plain_rows] = _per_query(index, plain_rows, ts[plain_rows], ti[plain_rows], presorted=True)
    if len(tie_rows):                                             # candidate lists longer than k: padded to the longest
        wmax = max(len(extra[int(r)][0]) for r in tie_rows)
        sc = np.full((len(tie_rows), wmax), -np.inf, dtype=np.float32)
        cols = np.full((len(tie_rows), wmax), -1, dtype=np.int64)
        for a, r in enumerate(tie_rows):
            c, s_ = extra[int(r)]
            sc[a, : len(c)], cols[a, : len(c)] = s_, c
        vals[:, tie_rows] = _per_query(index, tie_rows, sc, cols, presorted=False)
    if len(rows):
        tot = np.cumsum(vals[:, rows], axis=1)[:, -1]             # sequential over queries, like `acc += v`
    else:
        tot = np.zeros((5, len(ks)))
    den = max(len(rows), 1)
    den_mrr = max(index.n_qrels, 1)
    names = (("NDCG", "NDCG@{}", den), ("mAP", "MAP@{}", den), ("Recall", "Recall@{}", den), ("Precision", "P@{}", den),
             ("mRR", "MRR@{}", den_mrr))
    return {name: {fmt.format(k): round(tot[a, j] / d, 5) for j, k in enumerate(ks)} for a, (name, fmt, d) in enumerate(names)}
