#!/usr/bin/env python3
"""Pin the train.log contract with the reference's OWN consumer: a run log written by this repo's logger (utils.get_logger /
log_json, the records driver.py emits) is parsed by the reference's summary_results.parse_train_log_best_metrics, and its
directory names by extract_mf_from_dirname / normalize_dataset_name.  The fixture (tests/golden/summary_contract.json) holds the
log text (this repo's output) and what the reference's functions returned for it -- data only.

Runs only in the build container (needs /root/reference; openpyxl is absent and only used for the xlsx writer, so it is stubbed)."""
import json
import os
import sys
import tempfile
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))
import summary_recipe as R  # noqa: E402   (tests/summary_recipe.py: writes the log with this repo's logger)


def import_reference_summary():
    for name in ["openpyxl", "openpyxl.styles", "openpyxl.utils", "openpyxl.worksheet", "openpyxl.worksheet.table"]:
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["openpyxl"].Workbook = object
    sys.modules["openpyxl.styles"].Font = sys.modules["openpyxl.styles"].Alignment = object
    sys.modules["openpyxl.utils"].get_column_letter = lambda i: "A"
    sys.modules["openpyxl.worksheet.table"].Table = sys.modules["openpyxl.worksheet.table"].TableStyleInfo = object
    sys.path.insert(0, "/root/reference")
    import summary_results
    return summary_results


def main():
    ref = import_reference_summary()
    out = {"cases": []}
    with tempfile.TemporaryDirectory() as tmp:
        for case in R.CASES:
            run_dir, lines = R.write_log(tmp, case)
            parsed = ref.parse_train_log_best_metrics(os.path.join(run_dir, "train.log"))
            setting, mf_dir, ds_dir = case["setting"], f"mf{case['mf']}", case["dataset"]
            out["cases"].append({"name": case["name"], "log_lines_without_time": lines, "parsed_by_reference": parsed,
                                 "mf_from_dirname": ref.extract_mf_from_dirname(mf_dir),
                                 "dataset_name": ref.normalize_dataset_name(ds_dir)})
    with open(os.path.join(HERE, "summary_contract.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(json.dumps(out["cases"], indent=1)[:1500])


if __name__ == "__main__":
    main()
