#!/bin/bash
# HBM traffic per kernel from two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE cannot share a pass) of a
# short bench.py run; prints the per-kernel means (KiB) and writes gpurun_out/pmc_traffic/{summary.txt,traffic.json}.
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/pmc_traffic; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $out/$c --output-format csv -- python $R/bench.py --no-cpu-baseline --no-strict-fp32 --no-extra-legs --steps 3 --warmup 1 > $out/bench_$c.json 2> $out/err_$c.log
done
python $R/tools/pmc_summary.py $out/FETCH_SIZE $out/WRITE_SIZE > $out/summary.txt
python - <<PY
import re, json
txt = open("$out/summary.txt").read()
blk = re.search(r"mlp_fwd_rr_kernel<2, true(?:, false)?>\(pln\S*\s+grid=1572864\n((?:\s+\w+.*\n)+)", txt)
vals = dict(re.findall(r"(\w+_SIZE)\s+n=\s*\d+\s+mean=([\d.e+]+)", blk.group(1)))
f, w = float(vals["FETCH_SIZE"]), float(vals["WRITE_SIZE"])
json.dump({"commit": "${COMMIT:-unknown}", "f16x3": {"kernel": "mlp_fwd_rr_kernel<2,true>", "rows_per_launch": 786432, "fetch_size_kib": f, "write_size_kib": w,
                     "bytes": int((2 * f + w) * 1024), "algorithmic_bytes": 786432 * 4844}}, open("$out/traffic.json", "w"), indent=1)
print(open("$out/traffic.json").read())
PY
rm -rf $out/FETCH_SIZE $out/WRITE_SIZE
cat $out/summary.txt | head -60
