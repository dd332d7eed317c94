"""gpurun_out/pmc_<tag>/pmc_summary.json (raw counter means of scratch/pmc.sh) -> profiles/<tag>_pmc_summary.json (derived figures)
and profiles/hbm_traffic.json (what bench.py replays as roofline.traffic).  usage: python scratch/pmc_post.py r02 [kernel symbol]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
raw = json.load(open(os.path.join(ROOT, "gpurun_out", f"pmc_{tag}", "pmc_summary.json")))
line = None
for f in ("fetch.json", "sq1.json"):
    for ln in open(os.path.join(ROOT, "gpurun_out", f"pmc_{tag}", f)):
        if ln.startswith("{"):
            line = json.loads(ln)
kernel = sys.argv[2] if len(sys.argv) > 2 else line["roofline"]["kernel"]
m = lambda k: raw[k]["mean"]
kns = raw["_kernel_ns_under_pmc"]
t_sq1 = kns["sq1_kernel_trace.csv"]["mean"] * 1e-9
clock = m("GRBM_GUI_ACTIVE") / 8 / t_sq1 / 1e9                      # GRBM_GUI_ACTIVE sums the 8 XCDs
busy = m("SQ_VALU_MFMA_BUSY_CYCLES") / 1024 / (m("GRBM_GUI_ACTIVE") / 8)  # per-SIMD busy cycles summed over 256 CUs x 4 SIMDs
rd, wr = 2 * m("FETCH_SIZE") * 1024, m("WRITE_SIZE") * 1024
wc = m("SQ_WAVE_CYCLES")
shares = {"active": m("SQ_ACTIVE_INST_ANY") / wc,
          "issue_stall": m("SQ_WAIT_INST_ANY") / wc, "parked": m("SQ_WAIT_ANY") / wc}
out = {"round": int("".join(ch for ch in tag[1:] if ch.isdigit())), "workload": f"bench.py --steps 1 --no-other-regimes ({line['config']['queries_per_step']} queries x {line['config']['pages']} pages), kernel {kernel}",
       "kernel_ms_under_pmc": t_sq1 * 1e3, "effective_clock_ghz": clock, "mfma_busy_frac": busy,
       "lds_bank_conflict_cycles": m("SQ_LDS_BANK_CONFLICT"), "hbm_bytes_per_launch": rd + wr, "hbm_read_bytes": rd, "hbm_write_bytes": wr,
       "derivation": "clock = GRBM_GUI_ACTIVE/8/time; busy = SQ_VALU_MFMA_BUSY_CYCLES/1024/(GRBM_GUI_ACTIVE/8); bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 "
                     "(FETCH_SIZE in KB, halved on gfx950 for 16-B/lane reads: MI355X_MICROARCH.md 'HBM')",
       "wave_time_shares": shares, "counters": raw}
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_summary.json"), "w"), indent=1)
traffic = {"queries": line["config"]["queries_per_step"], "pages_per_gpu": line["config"]["pages"], "hbm_bytes_per_launch": rd + wr,
           "method": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (profiles/{tag}_pmc_summary.json, scratch/pmc.sh); FETCH_SIZE is in KB and reads 1/2 of "
                     "wide 16-B/lane streaming reads on gfx950 (MI355X_MICROARCH.md 'HBM'), so bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024; Infinity-Cache hits are included "
                     "in these fabric-side counters",
           "fetch_size_kb": m("FETCH_SIZE"), "write_size_kb": m("WRITE_SIZE"), "round": int("".join(ch for ch in tag[1:] if ch.isdigit())), "kernel": kernel}
json.dump(traffic, open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "counters"}, indent=1))
