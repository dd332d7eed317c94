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
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np


def _order_trec(docids: np.ndarray, scores: np.ndarray) -> np.ndarray:
    # primary: score desc; secondary: docid desc.  lexsort sorts ascending by the LAST key first.
    rank_of_id = np.empty(len(docids), dtype=np.int64)
    rank_of_id[np.argsort(docids, kind="stable")] = np.arange(len(docids))
    return np.lexsort((-rank_of_id, -scores))


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
            dcg = float((g * disc[: len(g)]).sum())
            ideal = pos_rels[:k]
            idcg = float((ideal * disc[: len(ideal)]).sum())
            acc["ndcg"][j] += dcg / idcg if idcg > 0 else 0.0
            h = hits[:k]
            nh = float(h.sum())
            acc["recall"][j] += nh / nrel if nrel else 0.0
            acc["p"][j] += nh / k
            if nrel and nh > 0:
                prec_at = np.cumsum(h) / np.arange(1, len(h) + 1)
                acc["map"][j] += float((prec_at * h).sum()) / nrel
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
