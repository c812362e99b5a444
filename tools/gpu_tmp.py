"""The sequence that exposed the two rules of DevBuf's mapping (engine.hip): one context, the window-record pass with int32, then one-byte,
then two-byte coverage -- the two-byte array is allocated right after the one-byte one is released -- each compared with the plain
int32 pass.  WIDTHS=4,1,2 (default) | 2 | 2,2 ..."""
import os, sys
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import numpy as np, torch
from raft_amd import engine, hostio
from raft_amd.params import RaftParams
from raft_amd.synth import make_overlaps
o = make_overlaps(412_500, mean_len=30000.0, coverage=32.0, seed=20241008, device="cuda:0")
p = RaftParams(est_cov=32)
e0 = engine.Engine(p, device=0)
e0.run_device(o.read_len, *o.columns()); s0 = e0.finish()
a = {k: v.clone() for k, v in e0.outputs_device().items()}
off = hostio.group_offsets(o.n_reads, o.qid.cpu().numpy())
win = hostio.pack_windows(o.qs.cpu().numpy(), o.qe.cpu().numpy(), p.reso)
B = int(((o.read_len.long() + 49) // 50).sum())
e1 = engine.Engine(RaftParams(est_cov=32, symmetric_mode=1), device=0)
d_off = torch.as_tensor(off).to("cuda:0"); d_w = torch.as_tensor(win.view(np.int32)).to("cuda:0")
for width in [int(x) for x in os.environ.get("WIDTHS", "4,1,2").split(",")]:
    e1.set_output_width(width)
    e1.run_device_windows(o.read_len, d_off, d_w, n_bins=B); s1 = e1.finish()
    b = e1.outputs_device()
    d = (a["cov"] != b["cov"]).nonzero().flatten()
    print("width", width, "mismatches", d.numel(), "of", B, "first", d[:8].tolist(), "last", d[-4:].tolist())
    if d.numel():
        i = int(d[0]); print("  a", a["cov"][i-2:i+6].tolist(), "b", b["cov"][i-2:i+6].tolist(), " idx/32MiB-in-u16:", i * 2 / (32 << 20), " idx*4/32MiB:", i * 4 / (32 << 20))
        dd = d.cpu().numpy(); runs = np.split(dd, np.where(np.diff(dd) != 1)[0] + 1); print("  runs", len(runs), [(int(r[0]), len(r)) for r in runs[:6]])
