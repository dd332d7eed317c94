"""The train.log contract, pinned by the reference's own consumer: tests/golden/summary_contract.json holds run logs written by
this repo's logger / driver record builders AND what the reference's summary_results.parse_train_log_best_metrics,
extract_mf_from_dirname and normalize_dataset_name returned for them (made in the build container by
tests/golden/make_golden_summary.py).  Here: today's logger still writes exactly those lines (time stamp aside), and the reference's
parse is the best-nDCG@5 record of the run, as a percentage -- so summary_results.py keeps working on this repo's output."""
import json
import os

import pytest

import summary_recipe as R

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = {c["name"]: c for c in json.load(open(os.path.join(HERE, "golden", "summary_contract.json")))["cases"]}


@pytest.mark.parametrize("case", R.CASES, ids=[c["name"] for c in R.CASES])
def test_run_log_is_what_the_reference_parser_was_fed(tmp_path, case):
    gold = GOLD[case["name"]]
    run_dir, lines = R.write_log(str(tmp_path), case)
    assert lines == gold["log_lines_without_time"]
    assert run_dir.endswith(os.path.join(case["setting"], f"mf{case['mf']}", case["dataset"]))      # <root>/<setting>/mf<k>/<dataset>
    # what the reference made of it: the run's best-nDCG@5 evaluation, fractions shown as percentages (summary_results.py:82-85)
    best = max(case["evals"], key=lambda e: (e[2], e[1]))
    scale = lambda v: v * 100.0 if 0.0 <= v <= 1.0 else v
    assert gold["parsed_by_reference"] == {"N@5": scale(best[2]), "R@1": scale(best[1])}
    assert gold["mf_from_dirname"] == case["mf"]
    assert gold["dataset_name"] == ("arxiv" if case["dataset"].lower() == "arxivqa" else case["dataset"].lower())
