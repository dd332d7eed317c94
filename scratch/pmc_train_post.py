"""gpurun_out/pmc_train_<tag>/pmc_train_raw.json (per-launch counter means of scratch/pmc_train.sh) -> profiles/<tag>_pmc_train.json
(derived figures per mode and kernel of the fused training step) and profiles/train_traffic.json (what bench_train.py replays as
`traffic` of each train_step.roofline entry, with the file's hash).  usage: python scratch/pmc_train_post.py r05"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", f"pmc_train_{tag}")
raw = json.load(open(os.path.join(src, "pmc_train_raw.json")))
B, N, LQ, LT, LS, D = 32, 500, 32, 1030, 206, 128
ALG = {   # algorithmic HBM bytes per launch
    "teacher": N * LT * D * 2 * 2,                       # two fp16 planes of the teacher pages, read once per query batch
    "student": N * LS * D * 2 * 2 + B * N * LQ * 2,      # two planes of the student pages + the argmax written
    "update": N * LS * D * 4 * 6 + N * LS * D * 2 * 2 + B * N * LQ * 2 + B * N * 4,   # x, exp_avg, exp_avg_sq read + written; planes written; argmax + g read
    "loss": B * N * 4 * 3,
}
out = {"round": int("".join(ch for ch in tag[1:] if ch.isdigit())),
       "workload": "bench_train.py --steps 20 --warmup 10 --only <mode> --no-roofline under rocprofv3 --pmc, one counter group per run "
                   "(scratch/pmc_train.sh); B=32, N=500, Lt=1030, Ls=206, fp32 as fp16 hi/lo planes; per-launch means over the last two "
                   "thirds of the launches of each kernel",
       "derivation": "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: FETCH_SIZE is in KB and reads 1/2 of wide 16-B/lane reads on gfx950 (global_load_dwordx4 and "
                     "LDS-DMA alike: every bulk read of these kernels), WRITE_SIZE exact for 16-B/lane stores (MI355X_MICROARCH.md 'HBM'); both are the L2's "
                     "FABRIC-side request counters: a request served by the 256-MiB Infinity Cache is COUNTED like one served by HBM, so residency in the "
                     "Infinity Cache cannot be read off them (only L2 hits are excluded); clock = GRBM_GUI_ACTIVE/8/time; "
                     "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES/1024/(GRBM_GUI_ACTIVE/8); l2_hit = TCC_HIT_sum/(TCC_HIT_sum+TCC_MISS_sum)",
       "kernels": raw["kernels"], "modes": {}}
for mode, roles in raw["counters"].items():
    mo = out["modes"][mode] = {}
    for role, c in roles.items():
        m = lambda k: c[k]["mean"] if k in c else None
        kns = c.get("_kernel_ns_under_pmc", {})
        rec = {"launches_averaged": next((v["launches"] for k, v in c.items() if k != "_kernel_ns_under_pmc"), None)}
        if m("FETCH_SIZE") is not None and m("WRITE_SIZE") is not None:
            rd, wr = 2 * m("FETCH_SIZE") * 1024, m("WRITE_SIZE") * 1024
            rec.update(fabric_read_bytes=rd, fabric_write_bytes=wr, fabric_bytes_per_launch=rd + wr,
                       algorithmic_bytes_per_launch=ALG.get(role), traffic_over_algorithmic=(rd + wr) / ALG[role] if role in ALG else None)
        t = None
        for k, v in kns.items():
            if k.endswith("_sq1"):
                t = v["mean"] * 1e-9
        if t is None and kns:
            t = sorted(v["mean"] for v in kns.values())[len(kns) // 2] * 1e-9
        rec["kernel_us_under_pmc"] = {k: v["mean"] * 1e-3 for k, v in kns.items()}
        if m("GRBM_GUI_ACTIVE") is not None and t:
            rec["effective_clock_ghz"] = m("GRBM_GUI_ACTIVE") / 8 / t / 1e9
            if m("SQ_VALU_MFMA_BUSY_CYCLES") is not None:
                rec["mfma_busy_frac"] = m("SQ_VALU_MFMA_BUSY_CYCLES") / 1024 / (m("GRBM_GUI_ACTIVE") / 8)
            wc = m("SQ_WAVE_CYCLES")
            if wc:
                rec["wave_time_shares"] = {"active": m("SQ_ACTIVE_INST_ANY") / wc, "issue_stall": m("SQ_WAIT_INST_ANY") / wc, "parked": m("SQ_WAIT_ANY") / wc}
        if m("SQ_LDS_BANK_CONFLICT") is not None:
            rec["lds_bank_conflict_cycles"] = m("SQ_LDS_BANK_CONFLICT")
            rec["lds_bank_conflict_over_lds_active"] = m("SQ_LDS_BANK_CONFLICT") / max(m("SQ_LDS_IDX_ACTIVE") or 1, 1)
            rec["insts_mfma"] = m("SQ_INSTS_MFMA")
            rec["insts_valu"] = m("SQ_INSTS_VALU")
            rec["insts_lds"] = m("SQ_INSTS_LDS")
        if m("TCC_HIT_sum") is not None:
            rec["l2_hit_frac"] = m("TCC_HIT_sum") / max(m("TCC_HIT_sum") + m("TCC_MISS_sum"), 1)
            rec["l2_requests"] = m("TCC_HIT_sum") + m("TCC_MISS_sum")
        rec["counters"] = {k: v for k, v in c.items() if k != "_kernel_ns_under_pmc"}
        mo[role] = rec
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_train.json"), "w"), indent=1)
fused = out["modes"].get("fused", {})
traffic = {"round": out["round"], "batch": B, "pages": N, "teacher_patches": LT, "student_patches": LS,
           "method": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over the fused step (profiles/{tag}_pmc_train.json, scratch/pmc_train.sh); "
                     "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 correction for 16-B/lane reads); fabric-side counters: Infinity-Cache hits are included",
           "kernels": {role: {"kernel": out["kernels"].get("fused", {}).get(role), "hbm_bytes_per_launch": r.get("fabric_bytes_per_launch"),
                              "algorithmic_bytes_per_launch": r.get("algorithmic_bytes_per_launch")} for role, r in fused.items()}}
json.dump(traffic, open(os.path.join(ROOT, "profiles", "train_traffic.json"), "w"), indent=1)
for mode, roles in out["modes"].items():
    for role, r in roles.items():
        print(mode, role, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items() if k not in ("counters", "kernel_us_under_pmc", "wave_time_shares")})
