#!/bin/bash
# End-of-round evidence (run on the GPU box via gpurun): the five rocprofv3 passes of scripts/profile_bench.sh for every tracked
# configuration with the round's last binary.  Usage: scripts/profile_round.sh <round tag, e.g. r05z>
T=${1:-r04z}
scripts/profile_bench.sh ${T}_cfg3 > /dev/null 2>&1
scripts/profile_bench.sh ${T}_cfg3_ragged --ragged > /dev/null 2>&1
scripts/profile_bench.sh ${T}_cfg3_fused --op fused > /dev/null 2>&1                   # reshape_and_cache + attention in one launch
scripts/profile_bench.sh ${T}_cfg3_newest --op newest > /dev/null 2>&1                 # append-read attention, no cache write (round 6)
scripts/profile_bench.sh ${T}_cfg3_newest_ragged --op newest --ragged > /dev/null 2>&1
scripts/profile_bench.sh ${T}_cfg3_fp8 --kv fp8 > /dev/null 2>&1
scripts/profile_bench.sh ${T}_cfg3_fp8_ragged --kv fp8 --ragged > /dev/null 2>&1
scripts/profile_bench.sh ${T}_cfg4 --config cfg4 > /dev/null 2>&1
scripts/profile_bench.sh ${T}_cfg4_ragged --config cfg4 --ragged > /dev/null 2>&1
scripts/profile_bench.sh ${T}_cfg2 --config cfg2 > /dev/null 2>&1
scripts/profile_bench.sh ${T}_cfg5_strong --config cfg5_strong > /dev/null 2>&1     # BASELINE configs[4] on one GPU (N = 1 of the strong curve)
scripts/profile_bench.sh ${T}_long_b1 --config long_b1 > /dev/null 2>&1             # batch 1 x 16384 tokens: a split kernel (workspace)
scripts/profile_bench.sh ${T}_long_b4 --config long_b4 > /dev/null 2>&1             # batch 4 x 8192 tokens
scripts/profile_bench.sh ${T}_long_gqa --config long_gqa > /dev/null 2>&1           # batch 4 x 8192 tokens, 32 / 8 heads x 128: four query heads per item
scripts/profile_bench.sh ${T}_long_32k --config long_32k > /dev/null 2>&1           # batch 48 x 32768 tokens: past the plain kernels' LDS, in rounds
for c in cfg3 cfg3_ragged cfg3_fused cfg3_newest cfg3_newest_ragged cfg3_fp8 cfg3_fp8_ragged cfg4 cfg4_ragged cfg2 cfg5_strong long_b1 long_b4 long_gqa long_32k; do
  python - "$T" "$c" <<'PY'
import json, sys, glob, shutil, os
t, c = sys.argv[1], sys.argv[2]
d = json.load(open(f"gpurun_out/prof_{t}_{c}/summary.json"))
pd = d.get("pa_v1_dispatches", {})
print(c, d.get("kernel_variant"), "mean_us %.2f" % pd.get("mean_us_after_warmup", 0), "median %.2f" % pd.get("median_us", 0),
      "bytes", d.get("algorithmic_bytes_per_launch"), "traffic", d.get("traffic_bytes_corrected", {}).get("total"))
for f in glob.glob(f"gpurun_out/prof_{t}_{c}/trace/**/*kernel_stats.csv", recursive=True):
    shutil.copy(f, f"gpurun_out/{t}_{c}_kernel_stats.csv")
shutil.copy(f"gpurun_out/prof_{t}_{c}/summary.json", f"gpurun_out/{t}_{c}_summary.json")
PY
done
rm -rf gpurun_out/prof_${T}_*/pmc_* gpurun_out/prof_${T}_*/trace
