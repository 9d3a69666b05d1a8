#!/bin/bash
# collect_round.sh <tag, e.g. round3>: every summary profiles/README.md lists for a round, on ONE GPU box, into gpurun_out/ (copy them to
# profiles/ afterwards).  Counters are collected in their own rocprofv3 runs (--kernel-trace + --pmc only), the program directly after "--".
set -e
tag=${1:-round3}
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
stats() {  # stats <name> <program args...>: rocprofv3 --kernel-trace --stats of a command, the kernel_stats csv -> gpurun_out/<tag>_<name>.csv
  local name=$1; shift
  rm -rf /tmp/ks_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$name -- python3 "$@" > $out/${tag}_$name.log 2>&1
  cp "$(ls /tmp/ks_$name/*/*kernel_stats.csv | head -1)" $out/${tag}_$name.csv
  echo "$name: $(wc -l < $out/${tag}_$name.csv) kernels"
}
stats kernel_stats_bench_steps3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-extra
stats kernel_stats_c3 $root/bench_psp.py --steps 3 --warmup 1 --skip-1024
stats kernel_stats_c5 $root/tools/bench_tools/psp_leg.py bf16 1024 4 0
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_bf16_$ctr
  rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pmc_bf16_$ctr --output-format csv -- python3 $root/tools/bench_tools/bf16_time.py > /dev/null 2>&1
done
python3 - "$root" "$tag" <<'PY'
import json, os, subprocess, sys
root, tag = sys.argv[1], sys.argv[2]
parse = os.path.join(root, "tools/bench_tools/pmc_parse.py")
res = {"note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on tools/bench_tools/bf16_time.py (bf16 convolution forward / adjoint / weight "
               "gradient on the StyleGAN2 decoder's big layers); mean_value = KB per launch; FETCH_SIZE is to be doubled on gfx950 (MI355X_MICROARCH.md, HBM section)"}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = json.loads(subprocess.check_output([sys.executable, parse, "/tmp/pmc_bf16_" + ctr]))
    res[ctr] = [r for r in rows if "bf16" in r["kernel"]]
json.dump(res, open(os.path.join(root, "gpurun_out", tag + "_pmc_bf16_conv.json"), "w"), indent=1)
print("pmc bf16 conv:", len(res["FETCH_SIZE"]), "kernels")
PY
cd $root
python3 bench.py --steps 10 --warmup 3 --profile-dump gpurun_out/${tag}_shapes.txt > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
python3 bench_psp.py > gpurun_out/${tag}_bench_psp.json 2> gpurun_out/${tag}_bench_psp.err
tail -c 600 gpurun_out/${tag}_bench.json
