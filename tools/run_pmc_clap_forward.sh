# usage: bash tools/run_pmc_clap_forward.sh <tag>   (GPU box) -> gpurun_out/clap_pmc_summary_<tag>.json
# HBM-side traffic of ONE HtsatEncoder.forward (512 clips), summed over all of its launches: FETCH_SIZE and WRITE_SIZE in separate
# --pmc passes, each at 2 and at 6 forwards; the difference / 4 drops everything that happens once (weight packing, the mel batch).
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  for N in 2 6; do
    timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmcf_${C}_${N}_$TAG -- python3 $R/tools/prof_clap_forward.py $N > $R/gpurun_out/pmcf_${C}_${N}_$TAG.log 2>&1
  done
done
python3 - <<PY
import csv, glob, json
def total(d, name):
    s = 0.0
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                s += float(r["Counter_Value"])
    return s
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    lo, hi = total("$R/gpurun_out/pmcf_%s_2_$TAG" % c, c), total("$R/gpurun_out/pmcf_%s_6_$TAG" % c, c)
    out[c] = {"launches": 4, "mean": (hi - lo) / 4.0, "how": "sum over every launch of HtsatEncoder.forward (512 clips): (total at 6 forwards - total at 2) / 4, KiB"}
json.dump(out, open("$R/gpurun_out/clap_pmc_summary_$TAG.json", "w"), indent=1)
print(json.dumps(out))
PY
rm -rf $R/gpurun_out/pmcf_*_$TAG
