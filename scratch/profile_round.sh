#!/bin/bash
# ONE driver for a round's GPU artefacts (replaces the per-round r0x_run*.sh / r0x_final*.sh one-offs).  Runs on the GPU box:
#   gpurun --timeout 1200 -- 'bash scratch/profile_round.sh r06 suite fuzz bench train'
# sections (any subset, in the order given):
#   suite      python -m pytest tests -m gpu (the whole suite; tests/test_gpu_sentinel.py runs it once more through libevdr_sentinel.so)
#   fuzz       one pass of scratch/fuzz_fwd.py / fuzz_bwd.py / fuzz_topk.py on the product library
#   bench      bench.py (the driver's command) + rocprofv3 --kernel-trace --stats of the same command
#   train      bench_train.py (all modes) + rocprofv3 traces of the fused and the cached step (+ exclusive per-kernel times)
#   pmc        scratch/pmc.sh + pmc_post.py: the headline kernel's counter passes (-> profiles/hbm_traffic.json, replayed by bench.py)
#   pmc_train  scratch/pmc_train.sh: counter passes of the three training kernels
#   sweep      scratch/small_nq.py: the 1..64-query regimes
# Everything lands in gpurun_out/<tag>_*; copy what is to be judged into profiles/.
set -o pipefail
TAG=${1:-r06}; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
stats() { ls /tmp/$1/*/*kernel_stats.csv /tmp/$1/*kernel_stats.csv 2>/dev/null | head -1; }
trace() { ls /tmp/$1/*/*kernel_trace.csv /tmp/$1/*kernel_trace.csv 2>/dev/null | head -1; }
for section in "$@"; do
  case $section in
    suite)
      (cd $R && timeout -k 10 1100 python3 -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/${TAG}_gputests_tail.txt); echo "suite rc=$?"; tail -3 $O/${TAG}_gputests_tail.txt ;;
    fuzz)
      (cd $R && {
        echo "# $TAG fuzz pass on the product library, one gpurun call"
        echo "## fuzz_fwd.py 96000 500"; timeout -k 10 400 python3 scratch/fuzz_fwd.py 96000 500 2>&1 | grep -v amdgpu.ids | tail -3
        echo "## fuzz_bwd.py 26000 400 (each seed also: two launches bit-equal)"; timeout -k 10 400 python3 scratch/fuzz_bwd.py 26000 400 2>&1 | grep -v amdgpu.ids | tail -3
        echo "## fuzz_topk.py 7000 200"; timeout -k 10 300 python3 scratch/fuzz_topk.py 7000 200 2>&1 | grep -v amdgpu.ids | tail -3
      } > $O/${TAG}_fuzz.txt 2>&1); echo "fuzz rc=$?"; cat $O/${TAG}_fuzz.txt ;;
    bench)
      cd /tmp
      python3 $R/bench.py > $O/${TAG}_bench_n1.json 2> $O/${TAG}_bench_n1.err; echo "bench rc=$?"
      rm -rf /tmp/prof_b; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o bench -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/${TAG}_prof_bench.json 2> $O/${TAG}_prof_bench.err; echo "prof bench rc=$?"
      cp $(stats prof_b) $O/${TAG}_bench_kernel_stats.csv
      head -c 400 $O/${TAG}_bench_n1.json; echo; python3 $R/scratch/kstats.py $O/${TAG}_bench_kernel_stats.csv 6 ;;
    train)
      cd /tmp
      python3 $R/bench_train.py --steps 60 > $O/${TAG}_bench_train.json 2> $O/${TAG}_bench_train.err; echo "train rc=$?"
      for mode in fused fused_cached; do
        rm -rf /tmp/prof_$mode; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$mode -o t -- python3 $R/bench_train.py --steps $([ $mode = fused ] && echo 30 || echo 60) --only $mode --no-cpu-baseline --no-roofline > $O/${TAG}_prof_train_$mode.json 2> $O/${TAG}_prof_train_$mode.err; echo "prof $mode rc=$?"
        cp $(stats prof_$mode) $O/${TAG}_train_${mode}_kernel_stats.csv
        # rocprofv3's intervals of back-to-back launches overlap; exclusive (queue-extending) time per kernel
        python3 $R/scratch/trace_exclusive.py --last $([ $mode = fused ] && echo 30 || echo 60) $(trace prof_$mode) "maxsim_fwd16s_kernel<2, 2, false" "maxsim_fwd16s_kernel<2, 2, true" maxsim_bwd_kernel infonce_row_kernel split_small_kernel split_segments_kernel vectorized_gather copyBuffer > $O/${TAG}_train_${mode}_trace_exclusive.json; echo "trace_exclusive $mode rc=$?"
      done
      cp $O/${TAG}_prof_train_fused.json $O/${TAG}_prof_train_line.json
      python3 -c "import json; r = json.load(open('$O/${TAG}_bench_train.json')); print({k: round(v['ms_per_step'], 4) for k, v in r['results'].items()})" ;;
    pmc)
      cd /tmp; bash $R/scratch/pmc.sh $TAG > $O/${TAG}_pmc.log 2>&1; echo "pmc rc=$?"
      (cd $R && python3 scratch/pmc_post.py $TAG > $O/${TAG}_pmc_post.log 2>&1); echo "pmc_post rc=$?" ;;
    pmc_train)
      cd /tmp; bash $R/scratch/pmc_train.sh $TAG > $O/${TAG}_pmc_train.log 2>&1; echo "pmc_train rc=$?" ;;
    sweep)
      (cd $R && python3 scratch/small_nq.py 40000 2>&1 | grep -v amdgpu.ids > $O/${TAG}_nq_sweep.txt); echo "sweep rc=$?"; cat $O/${TAG}_nq_sweep.txt ;;
    *) echo "unknown section $section" >&2; exit 2 ;;
  esac
done
