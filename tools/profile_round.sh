#!/bin/bash
# Produces the per-round evidence kept under profiles/: bench line, rocprofv3 kernel stats, PMC traffic, PCIe rate.
# usage: tools/profile_round.sh <tag> [bench args, e.g. --workload ultralong]      (run on the GPU box through gpurun)
TAG=${1:-r03_a}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
python bench.py "$@" > gpurun_out/$TAG/bench.log 2>&1; echo "bench rc=$?"; grep '^{' gpurun_out/$TAG/bench.log > gpurun_out/$TAG/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/stats -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-e2e --no-six-column-leg --no-packed-leg --no-placement-ab "$@" > gpurun_out/$TAG/stats.log 2>&1; echo "stats rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/$TAG/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-six-column-leg --no-packed-leg --no-placement-ab "$@" > gpurun_out/$TAG/pmc_fetch.log 2>&1; echo "pmc fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/$TAG/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-six-column-leg --no-packed-leg --no-placement-ab "$@" > gpurun_out/$TAG/pmc_write.log 2>&1; echo "pmc write rc=$?"
python3 - <<PY
import csv, glob, json, collections
tag = "$TAG"
def pileup_mean(d, counter):
    # per pass: pileup_wave_kernel (+ pileup_deep_kernel, which finds its list empty on the bench sets): means per launch, summed
    v = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/{tag}/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "pileup" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                v[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return sum(sum(x) / len(x) for x in v.values()) if v else None
fetch, write = pileup_mean("pmc_fetch", "FETCH_SIZE"), pileup_mean("pmc_write", "WRITE_SIZE")
print("FETCH_SIZE", fetch, "WRITE_SIZE", write)
b = json.load(open(f"gpurun_out/{tag}/bench.json"))
if fetch is not None and write is not None:
    # MI355X_MICROARCH.md §HBM: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports 1/2 of a wide coalesced read stream
    traffic = (2 * fetch + write) * 1024
    json.dump({"hbm_bytes_per_launch": traffic, "fetch_size_kib": fetch, "write_size_kib": write,
               "records_per_gpu": b["config"]["records_per_gpu"], "kernel_source_hash": b["roofline"]["kernel_source_hash"],
               "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), the pileup kernels of a pass (pileup_wave_kernel + the empty pileup_deep_kernel), means per launch summed; "
                         f"bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per MI355X_MICROARCH.md HBM section; session {tag} (gpurun_out/{tag}/pmc_traffic.json, kept as profiles/{tag}_pmc_traffic.json)",
               "session": tag},
              open(f"gpurun_out/{tag}/pmc_traffic.json", "w"), indent=1)
    print("traffic bytes", traffic, "algorithmic", b["roofline"]["bytes_algorithmic"])
PY
