#!/usr/bin/env python3
"""Prints the numbers DESIGN.md Part I quotes from an evidence directory (tools/evidence_round.sh <tag> -> gpurun_out/<tag> or profiles/<tag>_*)."""
import json, sys, os, glob
tag = sys.argv[1]
def L(name):
    for f in (f"gpurun_out/{tag}/{name}", f"profiles/{tag}_{name}"):
        if os.path.exists(f):
            for line in open(f):
                if line.startswith("{"):
                    return json.loads(line)
    return None
d = L("bench.json"); r = d["roofline"]
print(f"headline ms/step {d['ms_per_step']:.3f}  rec/s {d['value']:.3e}  frag/s {d['fragments_per_s']:.3e}  kernel {r['kernel_ms']:.3f} ms = {r['achieved']:.0f} GB/s frac {r['frac']:.3f}  pass {r['pass_device_ms']:.3f} frac {r['pass_frac']:.3f}  without cuts {r.get('pass_device_ms_without_cuts', 0):.3f}  traffic {r.get('traffic')}")
for k in ("grouped", "packed_output", "window_records", "window_records_delta4"):
    v = d.get(k)
    if v: print(f"  leg {k}: kernel {v.get('kernel_ms', 0):.3f} pass {v.get('pass_device_ms', 0):.3f} ms/step {v.get('ms_per_step', 0):.3f}" + (f" inspect-first six-column {v['six_column_pass_device_ms_inspect_first']:.3f}" if "six_column_pass_device_ms_inspect_first" in v else ""))
e = d.get("e2e")
if e: print(f"  e2e from_soa {e['records_per_s']:.3e} rec/s ({e['seconds']*1e3:.1f} ms) first pass {e['first_pass_s']:.3f} s; prepared {e.get('prepared_input', {}).get('records_per_s', 0):.3e} ({e.get('prepared_input', {}).get('seconds', 0)*1e3:.1f} ms); 12 B/record uploaded {e['six_column_input']['records_per_s']:.3e}; coordinate columns grouped {e['coordinate_columns']['records_per_s']:.3e}")
c = d.get("cpu_baseline")
if c: print(f"  cpu_baseline {c['value']:.3e} rec/s ({c['kind']})")
for name in ("bench_ultralong.json", "bench_s50k.json", "bench_slice412k.json", "bench_slice412k_windows_w1.json", "bench_shuffle.json", "bench_nonsym.json", "bench_variant0.json"):
    x = L(name)
    if not x: continue
    rr = x["roofline"]
    legs = {k: (round(v.get("kernel_ms", 0), 3), round(v.get("pass_device_ms", 0), 3)) for k, v in x.items() if isinstance(v, dict) and "pass_device_ms" in v and k != "roofline"}
    ee = x.get("e2e", {})
    print(f"{name}: ms/step {x['ms_per_step']:.3f} kernel {rr['kernel_ms']:.3f} pass {rr['pass_device_ms']:.3f} e2e {ee.get('records_per_s', 0):.3e} ({ee.get('seconds', 0)*1e3:.2f} ms) legs {legs}")
for sub in ("win", "ul"):
    f = f"gpurun_out/{tag}_{sub}/bench.json"
    if os.path.exists(f):
        x = [json.loads(l) for l in open(f) if l.startswith("{")][0]; rr = x["roofline"]
        print(f"{sub} (own run under rocprofv3 stats): ms/step {x['ms_per_step']:.3f} kernel {rr['kernel_ms']:.3f} pass {rr['pass_device_ms']:.3f}")
    f = f"gpurun_out/{tag}_{sub}/pmc_traffic.json"
    if os.path.exists(f): print(f"  {sub} traffic {json.load(open(f))['hbm_bytes_per_launch']/1e9:.3f} GB")
f = f"gpurun_out/{tag}/pmc_traffic.json"
if os.path.exists(f): t = json.load(open(f)); print(f"traffic {t['hbm_bytes_per_launch']/1e9:.3f} GB (fetch {t['fetch_size_kib']*2*1024/1e9:.3f} incl. x2, write {t['write_size_kib']*1024/1e9:.3f}) = {t['hbm_bytes_per_launch']/11433312200:.3f}x")
